// Software-pipelined HGEMM kernel for the large compute-bound tiles: ONE wave per SIMD.
//
// Measured on MI355X (profiles/): the 8-wave kernels (hgemm_tn_kernel and the two
// staggered variants in hgemm_kernel_pp.hpp) all stop at ~51 % MFMA-pipe utilisation on 4096^3
// regardless of schedule; the common limiter is the global->LDS path feeding 192 KiB of LDS
// fragment reads + 64 KiB of DMA per K-step per CU.  This family changes the geometry instead:
//   * 4 waves, one per SIMD, each owning the SIMD's whole 512-entry register file
//     (256 accumulator registers + two fragment sets), wave tile 128 x 128:
//     LDS fragment traffic per K-step drops from 192 KiB to 128 KiB and a barrier joins 4 waves.
//   * same LDS image and LDS-DMA staging as hgemm_tn_kernel (full 128-B lines, 2 stages);
//   * register-level software pipelining ACROSS the barriers: the fragments of the next K=32
//     slice are read behind the first MFMAs of the current slice, so the wave that owns the matrix
//     pipe never waits on LDS right after a barrier.
//   * an operand-split LDS-DMA stream with ~one K-step of flight time for EVERY piece.  Measured
//     on MI355X (profiles/): hipBLASLt's kernel of the same geometry has the same L2 hit rate but
//     parks its waves 10 % of the time against 36-43 % here; disassembling its main loop shows why:
//     it needs the A half of the next tile mid-loop and the B half later, and issues them in that
//     order a full iteration earlier.  Same idea here.  K-step t (stage s = t & 1) has 2*T MFMA
//     slots (T = FM*FN; interval A = slice 0, interval B = slice 1):
//
//       A[0, FM+FN)      ds_read slice 1 of tile t (A fragments, then B fragments) -> set B
//       X1  A[FM+FN+4]   lgkmcnt(0) + barrier: the A region of stage s is free
//       A(X1, T)         DMA: the A pieces of tile t+2 -> stage s
//       Y1  end of A     vmcnt + barrier: the A pieces of tile t+1 have landed (their B pieces and the
//                        A pieces of t+2 may still fly); also frees the B region of stage s
//       B[0, FM)         ds_read slice 0 A fragments of tile t+1 -> set A
//       B[2, T/2)        DMA: the B pieces of tile t+2 -> stage s (early: the latest piece sets the wait)
//       Y2  B[T/2]       vmcnt + barrier: the B pieces of tile t+1 have landed
//       B[T/2, T/2+FN)   ds_read slice 0 B fragments of tile t+1 -> set A
//
//     A pieces fly from A(X1,T) of K-step t to Y1 of K-step t+1, B pieces from interval B to Y2 of
//     the next K-step: >= ~100 MFMA slots (~1600 cycles) each, against ~48 with a single sync point.
//   * persistent: at most one workgroup per CU is launched and walks its work items ((split, tile)
//     pairs, XCD-local round-robin, see persistent_walk).  The LDS-DMA issue stream runs two K-steps
//     ahead of the MFMAs ACROSS work items, so the first two K-steps of the next output tile are in
//     flight while the current tile is converted and stored: no workgroup launch and no exposed
//     prologue latency per tile (measured: +12..20 % on K <= 1024 with M, N >= 8192, +2 % on 8192^3).
//   * two MFMA shapes: MI = 16 (v_mfma_f32_16x16x32_f16, 64 MFMAs per interval) and MI = 32
//     (v_mfma_f32_32x32x16_f16: two K=16 slices per interval, 32 MFMAs of 32 cycles).  Same LDS image, same
//     fragment bytes and the same slot plan in units of MFMA slots; the 32x32 form has half the MFMA issues
//     (its micro-benchmark ceiling is 11 % higher, MI355X_MICROARCH.md) and twice the issue room per slot for
//     the ds_read / LDS-DMA instructions that ride between the MFMAs.
#pragma once

#include "hgemm_kernel.hpp"

// DMA issue windows of the slot plan in 64ths of an interval (see SpPlan::AEND / BEND).  Measured: B pieces
// issued in the first half of interval B (33) instead of across all of it (61): 8192^3 883 -> 862 us.
#ifndef HGEMM_SP_AEND_64
#define HGEMM_SP_AEND_64 63
#endif
#ifndef HGEMM_SP_BEND_64
#define HGEMM_SP_BEND_64 33
#endif

namespace hgemm_mi355x {

// geometry = Cfg<BM, BN, WM, WN, MI, 2>; DMA pieces issued per MFMA group are derived below.
// An interval covers K = 32 of the stage: one 16x16x32 slice or two 32x32x16 slices.
template <int BM_, int BN_, int WM_, int WN_, int MI_ = 16>
struct CfgSP : Cfg<BM_, BN_, WM_, WN_, MI_, 2> {
  using Base = Cfg<BM_, BN_, WM_, WN_, MI_, 2>;
  static constexpr int SL  = (MI_ == 16) ? 1 : 2;          // MFMA k-slices per interval
  static constexpr int NFA = Base::FM * SL;                // A / B fragment reads (ds_read_b128) per interval
  static constexpr int NFB = Base::FN * SL;
  static constexpr int T   = Base::FM * Base::FN * SL;     // MFMA slots per interval
  static constexpr int ACC = (MI_ == 16) ? 4 : 16;         // accumulator registers per MFMA tile
  static constexpr int FLAG_OFF = Base::LDS_BYTES;         // LDS word of the fused split-K vote (behind the stages)
  static_assert(Base::NI % Base::NW == 0, "even DMA piece split");
  static_assert(T >= Base::NJ, "one MFMA slot per interleaved DMA piece");
  static_assert(Base::FM * Base::FN * ACC <= 256, "accumulators live in a0..a255");
};

#if defined(__HIP_DEVICE_COMPILE__)
// The accumulators never exist as C++ values: tile (i, j) of the wave tile lives in a[4n .. 4n+3],
// n = i*FN + j, named explicitly in every asm statement ("n" operands print as plain integers).
// hipcc otherwise (a) migrates fragments/accumulators between the VGPR and AGPR halves at 256
// accumulators per lane (v_accvgpr_read/write around every MFMA), and (b) copies ALL of them into
// VGPRs in one sweep in front of the epilogue, which in a persistent kernel spills everything that is
// live across the epilogue (next tile's fragments, DMA offsets).  sp_reserve_agprs() makes the kernel
// descriptor allocate a0..a255; nothing else may touch AGPRs: the kernel keeps its VGPR count far
// below 256 so the allocator never spills into them, and tests/test_build_audit.py checks that the
// ISA has no v_accvgpr_* outside the asm statements below.
// asm volatile also pins the issue ORDER, so the ds_read / LDS-DMA instructions written between two
// MFMAs stay between them.  No s_nop: with one wave per SIMD the 16-cycle MFMA issue interval leaves
// ~3 issue slots, and the fragment operands are only ever written by ds_read (waited for by
// lgkmcnt), never by a VALU instruction.
// N = 256 for every member that runs one workgroup per CU.  Round 5: a member whose LDS footprint lets TWO workgroups share a CU
// reserves exactly its accumulators, so that its waves stay within 256 registers and two of them fit a SIMD (round 3 launched
// 512 workgroups of q128x128 with all 256 AGPRs reserved -- 356 registers per wave, so only one workgroup per CU was ever
// resident and the other half queued: its "two workgroups per CU lose" figure measured that, not co-residency).
template <int N = 256>
__device__ __forceinline__ void sp_reserve_agprs() {
  static_assert(N == 256 || N == 128 || N == 96 || N == 64, "accumulator count of a two-resident member");
  if constexpr (N == 256) asm volatile("" ::: "a0", "a63", "a127", "a128", "a191", "a255");
  else if constexpr (N == 128) asm volatile("" ::: "a0", "a63", "a127");
  else if constexpr (N == 96) asm volatile("" ::: "a0", "a63", "a95");
  else asm volatile("" ::: "a0", "a63");
}
__device__ __forceinline__ void sp_mfma(int n, const f16x8& a, const f16x8& b) {
  asm volatile("v_mfma_f32_16x16x32_f16 a[%2:%3], %0, %1, a[%2:%3]" ::"v"(a), "v"(b), "n"(n * 4), "n"(n * 4 + 3));
}
__device__ __forceinline__ void sp_mfma32(int n, const f16x8& a, const f16x8& b) {
  asm volatile("v_mfma_f32_32x32x16_f16 a[%2:%3], %0, %1, a[%2:%3]" ::"v"(a), "v"(b), "n"(n * 16), "n"(n * 16 + 15));
}
template <int MI>
__device__ __forceinline__ void sp_mfma_mi(int n, const f16x8& a, const f16x8& b) {
  if constexpr (MI == 16) sp_mfma(n, a, b); else sp_mfma32(n, a, b);
}
// MFMA results -> v_accvgpr_read need the XDL write-back wait states; the compiler cannot see the
// dependency, so the epilogue opens with them explicitly (once per output tile).
__device__ __forceinline__ void sp_mfma_drain() { asm volatile("s_nop 15\n\ts_nop 15"); }
__device__ __forceinline__ f32x4 sp_read_acc(int n) {
  f32x4 r;
  asm volatile("v_accvgpr_read_b32 %0, a[%4]\n\tv_accvgpr_read_b32 %1, a[%5]\n\t"
               "v_accvgpr_read_b32 %2, a[%6]\n\tv_accvgpr_read_b32 %3, a[%7]"
               : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3])
               : "n"(n * 4), "n"(n * 4 + 1), "n"(n * 4 + 2), "n"(n * 4 + 3));
  return r;
}
__device__ __forceinline__ void sp_zero_acc(int n) {
  asm volatile("v_accvgpr_write_b32 a[%0], 0\n\tv_accvgpr_write_b32 a[%1], 0\n\t"
               "v_accvgpr_write_b32 a[%2], 0\n\tv_accvgpr_write_b32 a[%3], 0" ::"n"(n * 4), "n"(n * 4 + 1), "n"(n * 4 + 2), "n"(n * 4 + 3));
}

// Slot plan of one K-step (see the header comment).
template <class CFG>
struct SpPlan {
  static constexpr int FM = CFG::NFA, FN = CFG::NFB, T = CFG::T;   // fragment reads per operand, MFMA slots
  static constexpr int NJA = CFG::NI_A / CFG::NW;            // A pieces per wave per tile
  static constexpr int NJB = CFG::NJ - NJA;                  // B pieces per wave per tile
  static constexpr int RS = 2;                               // one fragment read every RS MFMA slots
  static constexpr int X1 = RS * FM + 4;                     // slot of interval A that carries the X1 sync
  static constexpr int Y2 = T / 2;                           // slot of interval B that carries the Y2 sync
  // last slot (exclusive) that may carry an A / B piece: issuing the pieces early in their interval
  // lengthens the flight time of the latest ones (they are the ones a sync point waits for)
  static constexpr int AEND = (T * HGEMM_SP_AEND_64) / 64 > X1 + NJA ? (T * HGEMM_SP_AEND_64) / 64 : X1 + NJA;
  static constexpr int BEND = (T * HGEMM_SP_BEND_64) / 64 > 2 + NJB ? (T * HGEMM_SP_BEND_64) / 64 : 2 + NJB;
  // A piece a (0..NJA-1) fires behind slot X1 + a*(T-1-X1)/NJA of interval A
  static constexpr int a_slot(int a) { return X1 + (a * (AEND - X1)) / NJA; }
  // B piece b (0..NJB-1) fires behind slot 2 + b*(T-3)/NJB of interval B
  static constexpr int b_slot(int b) { return 2 + (b * (BEND - 2)) / NJB; }
  static constexpr int a_at(int slot) { for (int a = 0; a < NJA; ++a) if (a_slot(a) == slot) return a; return -1; }
  static constexpr int b_at(int slot) { for (int b = 0; b < NJB; ++b) if (b_slot(b) == slot) return b; return -1; }
  static constexpr int b_before_y2() { int c = 0; for (int b = 0; b < NJB; ++b) c += b_slot(b) < Y2; return c; }
  static constexpr int NB1 = b_before_y2();                  // B pieces of tile t+2 issued ahead of Y2
  static_assert(CFG::NI_A % CFG::NW == 0 && CFG::NJ > NJA, "every wave owns whole A and B pieces");
  static_assert(X1 < T - NJA && RS * FM <= Y2 && Y2 + RS * FN <= T && RS * (FM + FN) <= T, "slot plan does not fit the interval");
  static_assert(a_slot(NJA - 1) < T && b_slot(NJB - 1) < T && a_slot(0) != a_slot(1 % NJA) , "distinct slots");
};

template <class CFG>
__device__ __forceinline__ void sp_issue_piece(__amdgpu_buffer_rsrc_t rs, const uint32_t (&voff)[CFG::NJ], char* lds_stage,
                                               int wave, int p, uint32_t kbyte) {
  lds_void_t* dst = (lds_void_t*)(lds_stage + (wave + p * CFG::NW) * 1024);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, dst, 16, voff[p], kbyte, 0, HGEMM_DMA_AUX);
}

__device__ __forceinline__ void sp_sync() {
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
}

// One interval (PHASE 0 = A, 1 = B): T MFMAs on (af, bf) with the reads / DMA pieces / sync points
// of the slot plan between them.  ONE body for all cases: two instantiations behind an if/else make
// hipcc shuffle all 256 accumulators.
template <class CFG, int PHASE>
__device__ __forceinline__ void sp_interval(const f16x8 (&af)[CFG::NFA],
                                            const f16x8 (&bf)[CFG::NFB], f16x8 (&naf)[CFG::NFA],
                                            f16x8 (&nbf)[CFG::NFB], const char* next_a0, const char* next_a1,
                                            const char* next_b0, const char* next_b1,
                                            __amdgpu_buffer_rsrc_t rsA, __amdgpu_buffer_rsrc_t rsB,
                                            const uint32_t (&voff)[CFG::NJ], int wave, char* stage2, uint32_t kbyte2) {
  using P = SpPlan<CFG>;
  constexpr int FM = CFG::FM, FN = CFG::FN, NFA = CFG::NFA, NFB = CFG::NFB, T = CFG::T, MI = CFG::MI;
  constexpr int RS = P::RS;
  // fragment read r of an operand: sub-slice u = r / F, row block r % F
  auto read_a = [&](int r) { return *(const f16x8*)((r / FM ? next_a1 : next_a0) + (r % FM) * MI * ROW_BYTES); };
  auto read_b = [&](int r) { return *(const f16x8*)((r / FN ? next_b1 : next_b0) + (r % FN) * MI * ROW_BYTES); };
#pragma unroll
  for (int n = 0; n < T; ++n) {
    const int u = n / (FM * FN), i = (n / FN) % FM, j = n % FN;   // k-slice, accumulator tile (i, j)
    if (PHASE == 0 && n == P::X1) {                       // X1: the slice-1 A-fragment reads of tile t are retired
      // LDS returns in order: only the B-fragment reads issued since may still be outstanding
      constexpr int YOUNGER = (P::X1 + RS - 1) / RS - NFA;
      asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(YOUNGER > 0 ? YOUNGER : 0) : "memory");
      sp_sync();
    }
    if (PHASE == 1 && n == P::Y2) {                       // Y2: B pieces of tile t+1 have landed
      wait_vmcnt<P::NJA + P::NB1>();
      sp_sync();
    }
    sp_mfma_mi<MI>(i * FN + j, bf[u * FN + j], af[u * FM + i]);
    // fragment reads of the next interval, one every RS slots so the four waves (which leave every
    // sync point together) do not saturate the LDS pipe (unconditional: behind the last tile they
    // read stale LDS, which is never used)
    if (n % RS == 0) {
      const int r = n / RS;
      if (PHASE == 0) {
        if (r < NFA) naf[r] = read_a(r);
        else if (r < NFA + NFB) nbf[r - NFA] = read_b(r - NFA);
      } else {
        if (r < NFA) naf[r] = read_a(r);
        else if (n >= P::Y2 && (n - P::Y2) / RS < NFB) nbf[(n - P::Y2) / RS] = read_b((n - P::Y2) / RS);
      }
    }
    // LDS-DMA pieces of tile t+2 into the stage tile t is vacating.  Branch-free: behind the last
    // tile kbyte2 is clamped to the last valid K offset, so the redundant pieces read valid memory
    // and land in a stage nobody reads again (one wave per SIMD: a branch per piece costs MFMA issue)
    if (PHASE == 0) {
      const int a = P::a_at(n);
      if (a >= 0) sp_issue_piece<CFG>(rsA, voff, stage2, wave, a, kbyte2);
    } else {
      const int b = P::b_at(n);
      if (b >= 0) sp_issue_piece<CFG>(rsB, voff, stage2, wave, P::NJA + b, kbyte2);
    }
  }
}
#endif  // __HIP_DEVICE_COMPILE__

// Per-lane LDS-DMA source offsets and (wave-uniform) descriptors of one work item's tile.  The
// descriptors must be PROVABLY uniform (SGPRs): hipcc otherwise wraps every LDS-DMA in a waterfall
// loop (v_readfirstlane / s_and_saveexec / s_cbranch_execnz), which shreds the MFMA stream -- hence
// plain locals (no struct passed by reference) and readfirstlane on the base pointers.
#define SP_LOAD_ISSUE_ITEM(ITEM)                                                                              \
  do {                                                                                                        \
    const TileCoord itc = map_logical(g, walk.base + walk.first + (ITEM) * walk.stride, BM, BN);              \
    const uintptr_t a_addr = reinterpret_cast<uintptr_t>(g.A + (size_t)itc.m0 * g.lda);                       \
    const uintptr_t b_addr = reinterpret_cast<uintptr_t>(g.Bt + (size_t)itc.n0 * g.ldb);                      \
    const uintptr_t a_uni = ((uintptr_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(a_addr >> 32)) << 32) | \
                            (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a_addr);                  \
    const uintptr_t b_uni = ((uintptr_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(b_addr >> 32)) << 32) | \
                            (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b_addr);                  \
    rsA = __builtin_amdgcn_make_buffer_rsrc((void*)a_uni, 0, 0xFFFFFFFFu, 0x00020000);                        \
    rsB = __builtin_amdgcn_make_buffer_rsrc((void*)b_uni, 0, 0xFFFFFFFFu, 0x00020000);                        \
    _Pragma("unroll") for (int j_ = 0; j_ < NJ; ++j_) {                                                       \
      const int i_ = wave + j_ * CFG::NW;                                                                     \
      const bool isA_ = i_ < CFG::NI_A;                                                                       \
      const int il_ = isA_ ? i_ : i_ - CFG::NI_A;                                                             \
      const int r_ = il_ * 8 + (lane >> 3);                                                                   \
      const int rmax_ = isA_ ? (g.M - 1 - itc.m0) : (g.N - 1 - itc.n0);                                       \
      const int ld_ = isA_ ? g.lda : g.ldb;                                                                   \
      const int chunk_ = (lane & 7) ^ (((il_ & 1) << 2) | (lane >> 4));                                       \
      voff[j_] = ((uint32_t)min(r_, rmax_) * (uint32_t)ld_ + (uint32_t)chunk_ * 8u) * 2u;                     \
    }                                                                                                         \
    iss_kbyte = (uint32_t)__builtin_amdgcn_readfirstlane(itc.k_begin * 2);                                    \
    iss_item = (ITEM); iss_kt = 0; iss_nk = __builtin_amdgcn_readfirstlane(itc.nk);                           \
  } while (0)

// Move the issue stream one K-step on; past the last step of the last item it stays put (the
// branch-free DMA then re-reads that valid tile into a stage nobody consumes).
#define SP_ISSUE_ADVANCE()                                   \
  do {                                                       \
    if (iss_kt + 1 < iss_nk) {                               \
      ++iss_kt;                                              \
      iss_kbyte += ROW_BYTES;                                \
    } else if (iss_item + 1 < walk.count) {                  \
      SP_LOAD_ISSUE_ITEM(iss_item + 1);                      \
    }                                                        \
  } while (0)

// One K-step: intervals A and B on stage (step & 1); the DMA pieces belong to stream step + 2.
#define SP_K_STEP()                                                                                            \
  do {                                                                                                         \
    char* st  = smem + (step & 1) * CFG::STAGE_BYTES;                                                          \
    char* nst = smem + ((step + 1) & 1) * CFG::STAGE_BYTES;                                                    \
    /* interval A: MFMAs on k 0..31 (set A); k 32..63 of this step streams into set B */                       \
    sp_interval<CFG, 0>(afA, bfA, afB, bfB, st + a_base_off + off10, st + a_base_off + off11,                  \
                        st + b_base_off + off10, st + b_base_off + off11, rsA, rsB, voff, wave, st, iss_kbyte); \
    /* Y1: my A pieces of step+1 have landed; its B pieces and the A pieces of step+2 may fly on.  The       \
       count is the number of YOUNGER LOADS only: loads retire in order among themselves, so it stays a      \
       safe (merely conservative) bound while an epilogue's stores are still in the VM queue.  Y1 also       \
       frees the B region of this stage for the B pieces issued in interval B, so every wave's fragment      \
       reads of it must have RETURNED (lgkmcnt), not merely been issued, before the barrier. */                \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                         \
    wait_vmcnt<P::NJB + P::NJA>();                                                                             \
    sp_sync();                                                                                                 \
    /* interval B: MFMAs on k 32..63 (set B); k 0..31 of step+1 streams into set A */                          \
    sp_interval<CFG, 1>(afB, bfB, afA, bfA, nst + a_base_off + off00, nst + a_base_off + off01,                \
                        nst + b_base_off + off00, nst + b_base_off + off01, rsA, rsB, voff, wave, st, iss_kbyte); \
    ++step;                                                                                                    \
  } while (0)

#if defined(__HIP_DEVICE_COMPILE__)
// ---- epilogue pieces.  A "unit" is what is live in VGPRs at once: one fragment row (MI = 16: FN tiles x 4
// registers) or one 32x32 tile (MI = 32: 16 registers).  Quad x = tile * NQ + q lives in a[4x .. 4x+3]. -----
// fp16 store of one 32x32 accumulator tile (operands swapped: lane holds C[m = lane & 31][n = 8q + 4(lane >> 5) + e],
// register 4q + e).  WIDE: v_permlane32_swap pairs quads (q, q+1) so every lane owns 8 consecutive N = one
// 16-byte store (cdna_hip_programming.md T21); otherwise four 8-byte stores.
template <int WIDE>
__device__ __forceinline__ void sp_store_tile32(const GemmArgs& g, int m, int nb, int lane, const f32x4 (&qd)[4]) {
  const bool wide = (WIDE == 1) || (WIDE < 0 && ((g.N & 7) == 0) && ((g.ldc & 7) == 0) &&
                                    ((reinterpret_cast<uintptr_t>(g.C) & 15) == 0));
  using h2 = __attribute__((ext_vector_type(2))) _Float16;
  if (wide) {
#pragma unroll
    for (int qp = 0; qp < 4; qp += 2) {
      const h2 a01 = {(f16)qd[qp][0], (f16)qd[qp][1]}, a23 = {(f16)qd[qp][2], (f16)qd[qp][3]};
      const h2 b01 = {(f16)qd[qp + 1][0], (f16)qd[qp + 1][1]}, b23 = {(f16)qd[qp + 1][2], (f16)qd[qp + 1][3]};
      // lanes 32-63 of the first operand <-> lanes 0-31 of the second: lower lanes end up with quad qp of
      // both halves (n = 8qp .. 8qp+7), upper lanes with quad qp+1 of both halves (n = 8(qp+1) .. +7)
      const auto r0 = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a01), __builtin_bit_cast(unsigned, b01), false, false);
      const auto r1 = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a23), __builtin_bit_cast(unsigned, b23), false, false);
      const int n = nb + 8 * (qp + (lane >> 5));
      if (m < g.M && n < g.N) {
        using u4 = __attribute__((ext_vector_type(4))) unsigned;
        const u4 o = {r0[0], r1[0], r0[1], r1[1]};
        HGEMM_STORE_C(g, (u4*)(g.C + (size_t)m * g.ldc + n), o);
      }
    }
  } else {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int n = nb + 8 * q + 4 * (lane >> 5);
      if (m < g.M && n < g.N) {
        const f16x4 o = {(f16)qd[q][0], (f16)qd[q][1], (f16)qd[q][2], (f16)qd[q][3]};
        HGEMM_STORE_C(g, (f16x4*)(g.C + (size_t)m * g.ldc + n), o);
      }
    }
  }
}
#endif  // __HIP_DEVICE_COMPILE__

// ---- LDS-staged fp16 epilogue (round 3) -----------------------------------------------------------------------------
// Timeline of the round-2 epilogue (tools/lab/gpu_round3_a.sh): ~8.3-9k cycles per 256x256 tile wherever it runs (alone at the
// end of a single-round grid or between the K loops of a persistent walk), i.e. ~16 B/clk/CU: it is bound by the store
// instruction, not by bandwidth.  In the MFMA layout a lane holds 4 (8 after the lane swap) consecutive N of ONE row, so a
// store instruction touches 16 rows x 64 B: sixteen half cache lines.  Here every wave turns its rows around in a private
// 4 KiB of LDS behind the pipeline stages instead: four 16x16 tiles (16 rows x 64 columns) are written as they sit in the
// accumulators (ds_write_b64, 16-B chunk c of row r at slot c ^ (r & 7)) and read back row-wise (ds_read_b128: lane ->
// row lane >> 3 (+ 8), chunk lane & 7), so that one store instruction covers 8 rows x 128 B = eight FULL lines.  Same
// number of store instructions, half the lines per instruction, no lane swaps.  Two 2 KiB buffers per wave alternate; DS
// instructions of one wave execute in order, so the only wait is for the read-back data.  The region is wave-private: no
// barrier, and the next work item's LDS-DMA stream (other regions) keeps flowing underneath.
#ifndef HGEMM_EPI_STAGED
#define HGEMM_EPI_STAGED 1
#endif
constexpr int SP_STAGED_BYTES_PER_WAVE = 4096;
// (only where the extra 16 KiB do not cost a resident workgroup: the host launches 256 x (160 KiB / (stages + 64 B)) of them)
template <class CFG>
constexpr bool sp_staged_ok(int lds_bytes) {
  return HGEMM_EPI_STAGED && CFG::MI == 16 && CFG::FN % 4 == 0 &&
         (160 * 1024) / (lds_bytes + 64 + CFG::NW * SP_STAGED_BYTES_PER_WAVE) == (160 * 1024) / (lds_bytes + 64) &&
         (160 * 1024) / (lds_bytes + 64) >= 1;
}
#if defined(__HIP_DEVICE_COMPILE__)
// whole wave tile (FM fragment rows x FN tiles) -> C; `stage` = this wave's 4 KiB.  Accumulators are re-zeroed behind the
// read when another work item follows.  The stores go through a buffer descriptor that starts at the wave tile's first
// element and ends with the matrix: rows past M are beyond its range and dropped by the hardware, lanes whose columns lie
// past N get an out-of-range offset; the position inside the wave tile is a scalar offset (no 64-bit pointer per store).
// (Reach: the host only takes this path when a 256-row tile spans less than 2 GiB of C, hgemm_api.hip.)
template <class CFG>
__device__ __forceinline__ void sp_epilogue_staged(const GemmArgs& g, int m_wave, int n_wave, bool rezero, char* stage) {
  constexpr int FM = CFG::FM, FN = CFG::FN, NG = FN / 4, GROUPS = FM * NG;
  using h2 = __attribute__((ext_vector_type(2))) _Float16;
  using u4 = __attribute__((ext_vector_type(4))) unsigned;
  // the lane id is re-derived HERE, opaquely: addresses computed from the kernel's own `lane` are loop-invariant, hipcc
  // hoists all of them to the kernel entry and they stay live across the K loop (+10 VGPRs, SGPR spills in the hot loop)
  int lane;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)stage;
  const int wrow = lane & 15, wq = lane >> 4;
  unsigned waddr[4];
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) waddr[jj] = lds0 + wrow * 128 + (((jj * 2 + (wq >> 1)) ^ (wrow & 7)) << 4) + (wq & 1) * 8;
  const int rrow = lane >> 3, rch = lane & 7;
  const unsigned raddr = lds0 + rrow * 128 + ((rch ^ (rrow & 7)) << 4);   // rows rrow and rrow + 8 (+1024: same swizzle key)
  // descriptor: base = &C[m_wave][n_wave] (wave-uniform, made provably so), range = the rest of the matrix
  const uintptr_t caddr = reinterpret_cast<uintptr_t>(g.C + (size_t)m_wave * g.ldc + n_wave);
  const uintptr_t cuni = ((uintptr_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(caddr >> 32)) << 32) |
                         (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)caddr);
  const long long rem = ((long long)(g.M - m_wave) * g.ldc - n_wave) * 2;              // bytes from the base to the end of C
  const unsigned range = __builtin_amdgcn_readfirstlane((int)(rem <= 0 ? 0 : rem > 0x7FFFFFFFLL ? 0x7FFFFFFF : rem));
  const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc((void*)cuni, 0, (int)range, 0x00020000);
  unsigned voff[NG];
#pragma unroll
  for (int gg = 0; gg < NG; ++gg)
    voff[gg] = (n_wave + gg * 64 + rch * 8 < g.N) ? (unsigned)(rrow * g.ldc + gg * 64 + rch * 8) * 2u : 0x80000000u;
  const unsigned row8 = (unsigned)g.ldc * 16u;                                          // 8 rows further, in bytes
  const bool nt = (g.flags & 1) != 0;                                                   // plan flag HGEMM_PLAN_NT_STORE (uniform)
  u4 rb[2][2];   // read-back data of the group in flight [buffer][row half]
  auto write_group = [&](int k) {      // accumulators of group k -> buffer k & 1
    const int i = k / NG, gg = k % NG;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int x = i * FN + gg * 4 + jj;
      const f32x4 v = sp_read_acc(x);
      if (rezero) sp_zero_acc(x);
      const h2 lo = {(f16)v[0], (f16)v[1]}, hi = {(f16)v[2], (f16)v[3]};
      const unsigned long long pk = (unsigned long long)__builtin_bit_cast(unsigned, lo) | ((unsigned long long)__builtin_bit_cast(unsigned, hi) << 32);
      if (k & 1) asm volatile("ds_write_b64 %0, %1 offset:2048" ::"v"(waddr[jj]), "v"(pk) : "memory");
      else       asm volatile("ds_write_b64 %0, %1" ::"v"(waddr[jj]), "v"(pk) : "memory");
    }
  };
  auto read_group = [&](int k) {
    if (k & 1) {
      asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(rb[1][0]) : "v"(raddr) : "memory");
      asm volatile("ds_read_b128 %0, %1 offset:3072" : "=v"(rb[1][1]) : "v"(raddr) : "memory");
    } else {
      asm volatile("ds_read_b128 %0, %1" : "=v"(rb[0][0]) : "v"(raddr) : "memory");
      asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(rb[0][1]) : "v"(raddr) : "memory");
    }
  };
  auto store_group = [&](int k, bool writes_behind) {
    const int i = k / NG, gg = k % NG, b = k & 1;
    // lgkmcnt(0), not a counted wait past the (up to four) ds_writes of the next group: LDS returns in order, but scalar loads
    // share the counter and return out of order -- a counted wait is only as good as the proof that no s_load is in the window
    // (rounds 3-4 audited the ISA for that instead; round 5: the writes have long retired when the read-back data arrives, the
    // plain wait costs nothing measurable, profiles/r05_*).
    (void)writes_behind;
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(rb[b][0]), "+v"(rb[b][1])::"memory");
    if (HGEMM_DBG(g, 2)) return;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (HGEMM_NT_STORE || nt) __builtin_amdgcn_raw_buffer_store_b128(rb[b][h], rsC, voff[gg], (unsigned)(i * 2 + h) * row8, 2);
      else                      __builtin_amdgcn_raw_buffer_store_b128(rb[b][h], rsC, voff[gg], (unsigned)(i * 2 + h) * row8, 0);
    }
  };
  write_group(0);
#pragma unroll
  for (int k = 0; k < GROUPS; ++k) {
    read_group(k);
    if (k + 1 < GROUPS) write_group(k + 1);
    store_group(k, k + 1 < GROUPS);
  }
}
#endif  // __HIP_DEVICE_COMPILE__

// EPI: 0 = narrow fp16 epilogue, 1 = wide fp16 epilogue (host checked N % 8, ldc % 8, 16-B aligned C),
//      2 = fp32 split-K partials for the two-pass combine / the hybrid tail, 3 = single-launch (fused) split-K
constexpr int SP_EPI_NARROW = 0, SP_EPI_WIDE = 1, SP_EPI_SLAB = 2, SP_EPI_FUSED = 3;

template <class CFG, int EPI>
__global__ void __launch_bounds__(CFG::THREADS) hgemm_tn_sp_kernel(const GemmArgs g) {
  prefetch_kernargs<sizeof(GemmArgs)>();
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int BM = CFG::BM, BN = CFG::BN, FM = CFG::FM, FN = CFG::FN, NJ = CFG::NJ, MI = CFG::MI;
  constexpr int NFA = CFG::NFA, NFB = CFG::NFB;
  constexpr int NQ = CFG::ACC / 4;              // f32x4 quads per accumulator tile

  // the two stages + one word for the fused split-K vote (ONE LDS object: a second one makes hipcc drain
  // vmcnt in front of every ds_read of the pipeline)
  constexpr bool STAGED = EPI == SP_EPI_WIDE && sp_staged_ok<CFG>(CFG::LDS_BYTES);   // + 4 KiB per wave for the epilogue
  __shared__ __attribute__((aligned(1024))) char smem[CFG::LDS_BYTES + 64 + (STAGED ? CFG::NW * SP_STAGED_BYTES_PER_WAVE : 0)];

  const int tid  = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave_m = wave / CFG::WN;
  const int wave_n = wave % CFG::WN;

  // persistent walk over this workgroup's work items ((split, tile) pairs)
  const ItemWalk walk = persistent_walk(g.items);
  if (walk.count == 0) return;

  // fragment offsets inside a stage (same image as hgemm_tn_kernel): interval h (K half), sub-slice u
  //   MI = 16: row lane & 15, 16-B chunk 4h + (lane >> 4);  MI = 32: row lane & 31, chunk 4h + 2u + (lane >> 5)
  const int lr = lane & (MI - 1), lq = lane / MI, sw = (lr >> 1) & 7;
  auto frag_off = [&](int h, int u) { return lr * ROW_BYTES + (((4 * h + 2 * u + lq) ^ sw) << 4); };
  const int off00 = frag_off(0, 0), off01 = frag_off(0, 1), off10 = frag_off(1, 0), off11 = frag_off(1, 1);
  const int a_base_off = wave_m * CFG::TM * ROW_BYTES;
  const int b_base_off = BM * ROW_BYTES + wave_n * CFG::TN * ROW_BYTES;

  sp_reserve_agprs();

  // ---- LDS-DMA issue stream: runs two K-steps ahead of the MFMAs and crosses work-item boundaries,
  //      so the first tiles of the next output tile are in flight while the current one is stored ----
  using P = SpPlan<CFG>;
  __amdgpu_buffer_rsrc_t rsA, rsB;
  uint32_t voff[NJ];
  uint32_t iss_kbyte;
  int iss_item, iss_kt, iss_nk;
  SP_LOAD_ISSUE_ITEM(0);
  // prologue: stream steps 0 and 1 in flight (A pieces first, then B pieces); step 0 visible
  stage_tile<CFG>(rsA, rsB, voff, smem, wave, iss_kbyte);
  SP_ISSUE_ADVANCE();
  stage_tile<CFG>(rsA, rsB, voff, smem + CFG::STAGE_BYTES, wave, iss_kbyte);
  SP_ISSUE_ADVANCE();
  // the accumulators are cleared behind the prologue's DMA issue, in the shadow of its HBM latency
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int x = 0; x < FM * FN * NQ; ++x) sp_zero_acc(x);
  wait_vmcnt<NJ>();
  __builtin_amdgcn_s_barrier();

  f16x8 afA[NFA], bfA[NFB], afB[NFA], bfB[NFB];
#pragma unroll
  for (int r = 0; r < NFA; ++r)
    afA[r] = *(const f16x8*)(smem + a_base_off + (r / FM ? off01 : off00) + (r % FM) * MI * ROW_BYTES);
#pragma unroll
  for (int r = 0; r < NFB; ++r)
    bfA[r] = *(const f16x8*)(smem + b_base_off + (r / FN ? off01 : off00) + (r % FN) * MI * ROW_BYTES);

  int step = 0;               // global K-step of this workgroup's stream: stage = step & 1
#pragma clang loop unroll(disable)
  for (int item = 0; item < walk.count; ++item) {
    const TileCoord tc = map_logical(g, walk.base + walk.first + item * walk.stride, BM, BN);
    const int nk = __builtin_amdgcn_readfirstlane(tc.nk);
    // hot loop: the issue stream stays inside this work item (steps t+2 and t+3 exist in it), so moving
    // it on is one scalar add; nothing but the slot plan lives in this loop
    int t = 0;
#pragma clang loop unroll(disable)
    for (; t + 3 < nk; ++t) {
      SP_K_STEP();
      iss_kbyte += ROW_BYTES;
      ++iss_kt;
    }
    // last (up to) three K-steps: the issue stream may cross into the next work item
#pragma clang loop unroll(disable)
    for (; t < nk; ++t) {
      SP_K_STEP();
      SP_ISSUE_ADVANCE();
    }
    // ---- epilogue of this work item; the next item's first two K-steps are already in flight.  Unit by
    //      unit: read the accumulators, re-zero them, convert, store -- one unit live in VGPRs ----------
    sp_mfma_drain();
    const bool rezero = item + 1 < walk.count;   // (the last tile's accumulators are not needed again)
    const int m_wave = tc.m0 + wave_m * CFG::TM, n_wave = tc.n0 + wave_n * CFG::TN;
    const __amdgpu_buffer_rsrc_t rsP = fused_rsrc(g);
    (void)m_wave; (void)n_wave; (void)rsP;
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
    } else {
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
      // single-launch split-K: publish my slab, draw a ticket; the last arriver of this tile adds the slabs
      // in split order and writes the fp16 tile (accumulators are not involved: they belong to the next item)
      if (fused_publish_and_vote(g, tc.tile, (volatile unsigned*)(smem + CFG::FLAG_OFF), tid)) {
        const int tiles = g.tiles_m * g.tiles_n;
        if constexpr (MI == 16) {
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
  }
  wait_vmcnt<0>();  // redundant tail pieces must not outlive the workgroup's LDS allocation
#endif  // __HIP_DEVICE_COMPILE__
}

#undef SP_LOAD_ISSUE_ITEM
#undef SP_ISSUE_ADVANCE
#undef SP_K_STEP

}  // namespace hgemm_mi355x
