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
//   * register-level software pipelining ACROSS the barrier: the fragments of the next K=32
//     slice are read while the current slice's 64 MFMAs run, so the wave that owns the matrix
//     pipe never waits on LDS right after a barrier.  Per K-step (tile t, stage s = t & 1):
//
//       interval A:  ds_read slice 1 of tile t  -> set B   ||  64 MFMAs on set A (slice 0)
//       lgkmcnt(0), vmcnt(0), s_barrier                        (tile t+1 landed; stage s is free)
//       interval B:  LDS-DMA tile t+2 -> stage s (interleaved) ||
//                    ds_read slice 0 of tile t+1 -> set A    ||  64 MFMAs on set B (slice 1)
//
//     One barrier per K-step.  At the barrier every wave has retired all of its reads of stage s
//     (WAR for the refill issued right after it) and has waited for its own pieces of tile t+1
//     (RAW for the reads issued right after it); only tile t+1 is ever outstanding at the wait,
//     so vmcnt(0) is exact, not a drain of younger loads.
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
// `s_nop 1`: wait states hipcc does not insert inside an asm string (VALU-written operand -> MFMA).
__device__ __forceinline__ void sp_mfma(f32x4& acc, const f16x8& a, const f16x8& b) {
  asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
// Last MFMA of an interval.  hipcc cannot see the MFMAs inside the asm statements, so it pads
// nothing between them and its own readers of their results -- and after the K loop it does place
// accumulator copies (v_accvgpr_read/mov from live-range splitting) straight behind the final MFMA;
// observed: element 0 of the last accumulator one K-slice stale.  The required MFMA-result ->
// reader wait states (8-pass XDL) therefore travel inside the statement itself: 16 states behind the
// last MFMA also cover its predecessors, which are at least one MFMA issue older each.
__device__ __forceinline__ void sp_mfma_last(f32x4& acc, const f16x8& a, const f16x8& b) {
  asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_f16 %0, %1, %2, %0\n\ts_nop 15" : "+a"(acc) : "v"(a), "v"(b));
}

// One interval: FM*FN MFMAs on the current fragment set (af, bf) with, in the issue slots between
// them, (a) the FM+FN fragment reads of the NEXT K=32 slice into (naf, nbf) when `prefetch`, and
// (b) this wave's NJ LDS-DMA pieces of the tile after next when `issue`.
template <class CFG>
__device__ __forceinline__ void sp_interval(f32x4 (&acc)[CFG::FM][CFG::FN], const f16x8 (&af)[CFG::FM],
                                            const f16x8 (&bf)[CFG::FN], f16x8 (&naf)[CFG::FM],
                                            f16x8 (&nbf)[CFG::FN], const char* next_a, const char* next_b,
                                            __amdgpu_buffer_rsrc_t rsA, __amdgpu_buffer_rsrc_t rsB,
                                            const uint32_t (&voff)[CFG::NJ], char* lds_stage, int wave,
                                            uint32_t kbyte, bool issue) {
  constexpr int FM = CFG::FM, FN = CFG::FN, TOTAL = FM * FN, NJ = CFG::NJ;
  constexpr int NRD = FM + FN;            // fragment reads of the next slice
  constexpr int DMA0 = NRD;               // DMA pieces go behind the reads
  static_assert(TOTAL >= NRD + NJ, "one issue slot per prefetch read and per DMA piece");
  constexpr int DSTEP = (TOTAL - DMA0) / NJ;
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int n = i * FN + j;
      if (n == TOTAL - 1) sp_mfma_last(acc[i][j], bf[j], af[i]);
      else sp_mfma(acc[i][j], bf[j], af[i]);
      if (n < NRD) {
        // unconditional: behind the last tile this reads stale (never used) LDS, no branch needed
        if (n < FN) nbf[n] = *(const f16x8*)(next_b + n * 16 * ROW_BYTES);
        else        naf[n - FN] = *(const f16x8*)(next_a + (n - FN) * 16 * ROW_BYTES);
      } else if ((n - DMA0) % DSTEP == 0 && (n - DMA0) / DSTEP < NJ) {
        if (issue) {  // wave-uniform; ONE body for both cases (two instantiations behind an if/else
                      // make hipcc shuffle all 256 accumulators between the branches)
          const int p = (n - DMA0) / DSTEP;
          const int piece = wave + p * CFG::NW;
          lds_void_t* dst = (lds_void_t*)(lds_stage + piece * 1024);
          if (piece < CFG::NI_A)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, dst, 16, voff[p], kbyte, 0, HGEMM_DMA_AUX);
          else
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, dst, 16, voff[p], kbyte, 0, HGEMM_DMA_AUX);
        }
      }
    }
}
#endif  // __HIP_DEVICE_COMPILE__

template <class CFG, bool SPLITK>
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

  // ---- prologue: tiles 0 and 1 in flight; tile 0 visible; slice 0 of tile 0 in set A -------------
  uint32_t kbyte = (uint32_t)tc.k_begin * 2u;
  stage_tile<CFG>(rsA, rsB, voff, smem, wave, kbyte);
  kbyte += ROW_BYTES;
  if (nk > 1) {
    stage_tile<CFG>(rsA, rsB, voff, smem + CFG::STAGE_BYTES, wave, kbyte);
    kbyte += ROW_BYTES;
    wait_vmcnt<NJ>();
  } else {
    wait_vmcnt<0>();
  }
  __builtin_amdgcn_s_barrier();

  f16x8 afA[FM], bfA[FN], afB[FM], bfB[FN];
#pragma unroll
  for (int j = 0; j < FN; ++j) bfA[j] = *(const f16x8*)(smem + b_base_off + off0 + j * 16 * ROW_BYTES);
#pragma unroll
  for (int i = 0; i < FM; ++i) afA[i] = *(const f16x8*)(smem + a_base_off + off0 + i * 16 * ROW_BYTES);

  for (int t = 0; t < nk; ++t) {
    char* st  = smem + (t & 1) * CFG::STAGE_BYTES;        // stage of tile t
    char* nst = smem + ((t + 1) & 1) * CFG::STAGE_BYTES;  // stage of tile t+1
    // ---- interval A: MFMAs on slice 0 (set A); slice 1 of tile t streams into set B ----------------
    sp_interval<CFG>(acc, afA, bfA, afB, bfB, st + a_base_off + off1, st + b_base_off + off1, rsA, rsB, voff,
                     st, wave, kbyte, false);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // my reads of stage (t&1) are retired
    wait_vmcnt<0>();                                      // my pieces of tile t+1 have landed
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // ---- interval B: MFMAs on slice 1 (set B); slice 0 of tile t+1 streams into set A; the refill
    //      of stage (t&1) with tile t+2 is issued behind the reads ------------------------------------
    const bool issue = (t + 2 < nk) && !(g.debug & 1);
    sp_interval<CFG>(acc, afB, bfB, afA, bfA, nst + a_base_off + off0, nst + b_base_off + off0, rsA, rsB, voff,
                     st, wave, kbyte, issue);
    if (t + 2 < nk) kbyte += ROW_BYTES;
  }
  store_tile<16, FM, FN, CFG::TM, CFG::TN, SPLITK>(g, tc, wave_m, wave_n, lane, acc);
#endif  // __HIP_DEVICE_COMPILE__
}

}  // namespace hgemm_mi355x
