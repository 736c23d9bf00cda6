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
//       B[2, T)          DMA: the B pieces of tile t+2 -> stage s
//       Y2  B[T/2]       vmcnt + barrier: the B pieces of tile t+1 have landed
//       B[T/2, T/2+FN)   ds_read slice 0 B fragments of tile t+1 -> set A
//
//     A pieces fly from A(X1,T) of K-step t to Y1 of K-step t+1, B pieces from interval B to Y2 of
//     the next K-step: >= ~100 MFMA slots (~1600 cycles) each, against ~48 with a single sync point.
#pragma once

#include "hgemm_kernel.hpp"

namespace hgemm_mi355x {

// geometry = Cfg<BM, BN, WM, WN, 16, 2>; DMA pieces issued per MFMA group are derived below
template <int BM_, int BN_, int WM_, int WN_>
struct CfgSP : Cfg<BM_, BN_, WM_, WN_, 16, 2> {
  using Base = Cfg<BM_, BN_, WM_, WN_, 16, 2>;
  static_assert(Base::NI % Base::NW == 0, "even DMA piece split");
  static_assert(Base::FM * Base::FN >= Base::NJ, "one MFMA slot per interleaved DMA piece");
};

#if defined(__HIP_DEVICE_COMPILE__)
// Accumulators are pinned to the accumulation half of the register file ("a" constraint) and the
// MFMA is issued from inline asm with dst tied to srcC.  hipcc's allocator otherwise migrates
// fragments/accumulators between the VGPR and AGPR halves at 256 accumulators per lane
// (v_accvgpr_read/write around every MFMA).  asm volatile also pins the issue ORDER, so the
// ds_read / LDS-DMA instructions written between two MFMAs below stay between them.
// No leading s_nop: with one wave per SIMD the 16-cycle MFMA issue interval leaves ~3 issue slots,
// and a nop per MFMA next to the interleaved ds_read / DMA instructions overflows them.  The
// fragment operands are only ever written by ds_read (waited for by lgkmcnt), never by a VALU
// instruction; tests/test_build_audit.py asserts that the loop contains no VALU write (v_mov etc.).
__device__ __forceinline__ void sp_mfma(f32x4& acc, const f16x8& a, const f16x8& b) {
  asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
// Last MFMA of an interval.  hipcc cannot see the MFMAs inside the asm statements, so it pads
// nothing between them and its own readers of their results -- and after the K loop it does place
// accumulator copies (v_accvgpr_read/mov from live-range splitting) straight behind the final MFMA;
// observed: element 0 of the last accumulator one K-slice stale.  The required MFMA-result ->
// reader wait states (8-pass XDL) therefore travel inside the statement itself: 16 states behind the
// last MFMA also cover its predecessors, which are at least one MFMA issue older each.
__device__ __forceinline__ void sp_mfma_last(f32x4& acc, const f16x8& a, const f16x8& b) {
  asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0\n\ts_nop 15" : "+a"(acc) : "v"(a), "v"(b));
}

// Slot plan of one K-step (see the header comment).
template <class CFG>
struct SpPlan {
  static constexpr int FM = CFG::FM, FN = CFG::FN, T = FM * FN;
  static constexpr int NJA = CFG::NI_A / CFG::NW;            // A pieces per wave per tile
  static constexpr int NJB = CFG::NJ - NJA;                  // B pieces per wave per tile
  static constexpr int RS = 2;                               // one fragment read every RS MFMA slots
  static constexpr int X1 = RS * FM + 4;                     // slot of interval A that carries the X1 sync
  static constexpr int Y2 = T / 2;                           // slot of interval B that carries the Y2 sync
  // A piece a (0..NJA-1) fires behind slot X1 + a*(T-1-X1)/NJA of interval A
  static constexpr int a_slot(int a) { return X1 + (a * (T - 1 - X1)) / NJA; }
  // B piece b (0..NJB-1) fires behind slot 2 + b*(T-3)/NJB of interval B
  static constexpr int b_slot(int b) { return 2 + (b * (T - 3)) / NJB; }
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
// of the slot plan between them.  `has1` / `has2` are wave-uniform (tile t+1 / t+2 exist).  ONE body
// for all cases: two instantiations behind an if/else make hipcc shuffle all 256 accumulators.
template <class CFG, int PHASE>
__device__ __forceinline__ void sp_interval(f32x4 (&acc)[CFG::FM][CFG::FN], const f16x8 (&af)[CFG::FM],
                                            const f16x8 (&bf)[CFG::FN], f16x8 (&naf)[CFG::FM],
                                            f16x8 (&nbf)[CFG::FN], const char* next_a, const char* next_b,
                                            __amdgpu_buffer_rsrc_t rsA, __amdgpu_buffer_rsrc_t rsB,
                                            const uint32_t (&voff)[CFG::NJ], int wave, char* stage2, uint32_t kbyte2,
                                            bool has1, bool has2) {
  using P = SpPlan<CFG>;
  constexpr int FM = CFG::FM, FN = CFG::FN, T = FM * FN;
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int n = i * FN + j;
      if (PHASE == 0 && n == P::X1) {                       // X1: the slice-1 A-fragment reads of tile t are retired
        // LDS returns in order: only the B-fragment reads issued since may still be outstanding
        constexpr int YOUNGER = (P::X1 + P::RS - 1) / P::RS - FM;
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(YOUNGER > 0 ? YOUNGER : 0) : "memory");
        sp_sync();
      }
      if (PHASE == 1 && n == P::Y2) {                       // Y2: B pieces of tile t+1 have landed
        wait_vmcnt<P::NJA + P::NB1>();
        sp_sync();
      }
      if (n == T - 1) sp_mfma_last(acc[i][j], bf[j], af[i]);
      else sp_mfma(acc[i][j], bf[j], af[i]);
      // fragment reads of the next slice, one every RS slots so the four waves (which leave every
      // sync point together) do not saturate the LDS pipe (unconditional: behind the last tile they
      // read stale LDS, which is never used)
      constexpr int RS = P::RS;
      if (n % RS == 0) {
        const int r = n / RS;
        if (PHASE == 0) {
          if (r < FM) naf[r] = *(const f16x8*)(next_a + r * 16 * ROW_BYTES);
          else if (r < FM + FN) nbf[r - FM] = *(const f16x8*)(next_b + (r - FM) * 16 * ROW_BYTES);
        } else {
          if (r < FM) naf[r] = *(const f16x8*)(next_a + r * 16 * ROW_BYTES);
          else if (n >= P::Y2 && (n - P::Y2) / RS < FN) nbf[(n - P::Y2) / RS] = *(const f16x8*)(next_b + ((n - P::Y2) / RS) * 16 * ROW_BYTES);
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
  (void)has1; (void)has2;
}
#endif  // __HIP_DEVICE_COMPILE__

// EPI: 0 = narrow fp16 epilogue, 1 = wide fp16 epilogue (host checked N % 8, ldc % 8, 16-B aligned C),
//      2 = fp32 split-K partials
template <class CFG, int EPI>
__global__ void __launch_bounds__(CFG::THREADS) hgemm_tn_sp_kernel(const GemmArgs g) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int BM = CFG::BM, BN = CFG::BN, FM = CFG::FM, FN = CFG::FN, NW = CFG::NW, NJ = CFG::NJ;

  __shared__ __attribute__((aligned(1024))) char smem[CFG::LDS_BYTES];

  const int tid  = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave_m = wave / CFG::WN;
  const int wave_n = wave % CFG::WN;

  const TileCoord tc = map_block(g, BM, BN);
  const int nk = tc.nk;

  const f16* a_base = g.A + (size_t)tc.m0 * g.lda;
  const f16* b_base = g.Bt + (size_t)tc.n0 * g.ldb;
  __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)a_base, 0, 0xFFFFFFFFu, 0x00020000);
  __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)b_base, 0, 0xFFFFFFFFu, 0x00020000);
  uint32_t voff[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int i    = wave + j * NW;
    const bool isA = i < CFG::NI_A;
    const int il   = isA ? i : i - CFG::NI_A;
    const int r    = il * 8 + (lane >> 3);
    const int rmax = isA ? (g.M - 1 - tc.m0) : (g.N - 1 - tc.n0);
    const int rc   = min(r, rmax);
    const int ld   = isA ? g.lda : g.ldb;
    const int chunk = (lane & 7) ^ (((il & 1) << 2) | (lane >> 4));
    voff[j] = ((uint32_t)rc * (uint32_t)ld + (uint32_t)chunk * 8u) * 2u;
  }

  // fragment offsets of the two K=32 slices inside a stage (same image as hgemm_tn_kernel)
  const int l15 = lane & 15, lq = lane >> 4, sw = l15 >> 1;
  const int off0 = l15 * ROW_BYTES + (((0 * 4 + lq) ^ sw) << 4);
  const int off1 = l15 * ROW_BYTES + (((1 * 4 + lq) ^ sw) << 4);
  const int a_base_off = wave_m * CFG::TM * ROW_BYTES;
  const int b_base_off = BM * ROW_BYTES + wave_n * CFG::TN * ROW_BYTES;

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- prologue: tiles 0 and 1 in flight (A pieces first, then B pieces); tile 0 visible --------
  using P = SpPlan<CFG>;
  const uint32_t kb0 = (uint32_t)tc.k_begin * 2u;
  stage_tile<CFG>(rsA, rsB, voff, smem, wave, kb0);
  // tile 1 (or, for a single-tile K range, tile 0 again: same branch-free clamping as in the loop)
  stage_tile<CFG>(rsA, rsB, voff, smem + CFG::STAGE_BYTES, wave, kb0 + (nk > 1 ? ROW_BYTES : 0));
  wait_vmcnt<NJ>();
  __builtin_amdgcn_s_barrier();

  f16x8 afA[FM], bfA[FN], afB[FM], bfB[FN];
#pragma unroll
  for (int i = 0; i < FM; ++i) afA[i] = *(const f16x8*)(smem + a_base_off + off0 + i * 16 * ROW_BYTES);
#pragma unroll
  for (int j = 0; j < FN; ++j) bfA[j] = *(const f16x8*)(smem + b_base_off + off0 + j * 16 * ROW_BYTES);

  for (int t = 0; t < nk; ++t) {
    char* st  = smem + (t & 1) * CFG::STAGE_BYTES;        // stage of tile t (and of tile t+2)
    char* nst = smem + ((t + 1) & 1) * CFG::STAGE_BYTES;  // stage of tile t+1
    // K byte offset of tile t+2, clamped to the last tile (see sp_interval: branch-free DMA)
    const uint32_t kb2 = kb0 + (uint32_t)min(t + 2, nk - 1) * ROW_BYTES;
    const bool has1 = (t + 1 < nk), has2 = (t + 2 < nk);
    // ---- interval A: MFMAs on slice 0 (set A); slice 1 of tile t streams into set B -----------------
    sp_interval<CFG, 0>(acc, afA, bfA, afB, bfB, st + a_base_off + off1, st + b_base_off + off1, rsA, rsB, voff, wave,
                        st, kb2, has1, has2);
    // Y1: my A pieces of tile t+1 have landed; its B pieces and the A pieces of tile t+2 may fly on
    wait_vmcnt<P::NJB + P::NJA>();
    sp_sync();
    // ---- interval B: MFMAs on slice 1 (set B); slice 0 of tile t+1 streams into set A ---------------
    sp_interval<CFG, 1>(acc, afB, bfB, afA, bfA, nst + a_base_off + off0, nst + b_base_off + off0, rsA, rsB, voff, wave,
                        st, kb2, has1, has2);
  }

  wait_vmcnt<0>();  // the redundant tail pieces must not outlive the workgroup's LDS allocation
  store_tile<16, FM, FN, CFG::TM, CFG::TN, EPI == 2, EPI == 2 ? -1 : EPI>(g, tc, wave_m, wave_n, lane, acc);
#endif  // __HIP_DEVICE_COMPILE__
}

}  // namespace hgemm_mi355x
