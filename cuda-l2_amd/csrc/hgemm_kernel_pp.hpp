// "Ping-pong" HGEMM kernel for the large compute-bound tiles (8 waves = two staggered groups).
//
// Why a second family: in hgemm_tn_kernel all 8 waves of a 256x256 workgroup hit the K-step
// barrier together, so the two waves that share a SIMD (and its matrix pipe) both sit in their
// LDS-read prologue at the same time and the MFMA pipe idles ~40 % of every K-step.  Here the
// workgroup is split into wave groups G0 = waves 0-3 and G1 = waves 4-7 (one wave of each group
// per SIMD) that run the same phase sequence ONE BARRIER APART:
//
//        interval:   I0      I1      I2      I3      I4     ...
//        G0:        R(0)    M(0)    R(1)    M(1)    R(2)          R(u) = ds_read fragments of
//        G1:         -      R(0)    M(0)    R(1)    M(1)                 half-tile u (K = 32)
//                                                                 M(u) = FM*FN MFMAs on them
//
// so in every interval one wave per SIMD feeds the matrix pipe while its partner reads LDS, and
// s_setprio(1) around M(u) lets the MFMA wave win issue arbitration.  (Same idea as the 8-phase
// schedule described in the CDNA4 programming guide; written from that description.)
//
// Data movement is at HALF-tile granularity (K = 32 halfs = 64 B per row):
//   * LDS = ring of 4 half-tile slots, each [(BM + BN) rows][64 B]; half-tile u lives in slot u%4.
//   * one LDS-DMA piece = 16 rows x 64 B = 1 KiB; chunk c (16 B) of row r is stored at slot
//     c ^ (((r >> 3) & 1) << 1) -- applied on the source address, undone on the fragment read --
//     which makes every ds_read_b128 lane group hit 16 distinct 16-B slots (conflict-free).
//   * each wave issues its P pieces of half-tile u+3 *inside* M(u), interleaved with the MFMAs
//     (the issue slots are free while the matrix pipe is busy), i.e. three half-tiles ahead.
//   * counted waits, never vmcnt(0) in steady state: before the barrier that ends interval
//     I(2u+1) half-tile u+1 must have landed:  G0 (after M(u), 2 younger half-tiles in flight)
//     waits vmcnt(2P);  G1 (after R(u), 1 younger half-tile in flight) waits vmcnt(P).
//   * hazards by construction: slot (u+3)%4 was last read in R(u-1): by G1 in I(2u-1), retired by
//     the lgkmcnt(0) ahead of its M(u-1) in I(2u); the earliest refill is issued in I(2u+1),
//     one barrier later.  A slot is read (I(2u+2) at the earliest) only after every wave waited
//     for its own pieces of it and passed the barrier ending I(2u+1).
#pragma once

#include "hgemm_kernel.hpp"

namespace hgemm_mi355x {

template <int BM_, int BN_, int WM_, int WN_, int MODE_ = 0>
struct CfgPP {
  static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_, MI = 16, NBUF = 4, MODE = MODE_;
  static constexpr int NW          = WM * WN;
  static constexpr int THREADS     = NW * 64;
  static constexpr int TM          = BM / WM;
  static constexpr int TN          = BN / WN;
  static constexpr int FM          = TM / 16;
  static constexpr int FN          = TN / 16;
  static constexpr int HROW_BYTES  = 64;                       // one half-tile row (32 halfs)
  static constexpr int HALF_BYTES  = (BM + BN) * HROW_BYTES;   // one ring slot
  static constexpr int LDS_BYTES   = HALF_BYTES * 4;
  static constexpr int NIH_A       = BM / 16;                  // 1-KiB pieces of the A half-tile
  static constexpr int NIH         = (BM + BN) / 16;
  static constexpr int P           = NIH / NW;                 // pieces per wave per half-tile
  static_assert(NW == 8, "two staggered groups of four waves");
  static_assert(BM % (WM * 16) == 0 && BN % (WN * 16) == 0, "wave tile must be MFMA-aligned");
  static_assert(NIH % NW == 0, "counted vmcnt needs an even piece split");
  static_assert(FM * FN >= P, "one MFMA slot per interleaved DMA piece");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};

#if defined(__HIP_DEVICE_COMPILE__)
template <class CFG>
__device__ __forceinline__ void pp_mfma_phase(f32x4 (&acc)[CFG::FM][CFG::FN], const f16x8 (&af)[CFG::FM],
                                              const f16x8 (&bf)[CFG::FN], __amdgpu_buffer_rsrc_t rsA,
                                              __amdgpu_buffer_rsrc_t rsB, const uint32_t (&voff)[CFG::P],
                                              char* lds_slot, int wave, uint32_t kbyte, bool issue) {
  constexpr int TOTAL = CFG::FM * CFG::FN;  // piece p is issued behind MFMA ((2p+1)*TOTAL)/(2P)
#pragma unroll
  for (int i = 0; i < CFG::FM; ++i)
#pragma unroll
    for (int j = 0; j < CFG::FN; ++j) {
      acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j], af[i], acc[i][j], 0, 0, 0);
      {
        const int n = i * CFG::FN + j;
        const int p = (n * 2 * CFG::P) / (2 * TOTAL);  // candidate piece for this slot (compile time)
        if (p < CFG::P && n == ((2 * p + 1) * TOTAL) / (2 * CFG::P) && issue) {  // `issue` is wave-uniform
          const int piece = wave + p * CFG::NW;  // wave-uniform
          lds_void_t* dst = (lds_void_t*)(lds_slot + piece * 1024);
          if (piece < CFG::NIH_A)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, dst, 16, voff[p], kbyte, 0, HGEMM_DMA_AUX);
          else
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, dst, 16, voff[p], kbyte, 0, HGEMM_DMA_AUX);
        }
      }
    }
}

template <class CFG>
__device__ __forceinline__ void pp_stage_half(__amdgpu_buffer_rsrc_t rsA, __amdgpu_buffer_rsrc_t rsB,
                                              const uint32_t (&voff)[CFG::P], char* lds_slot, int wave,
                                              uint32_t kbyte) {
#pragma unroll
  for (int p = 0; p < CFG::P; ++p) {
    const int piece = wave + p * CFG::NW;
    lds_void_t* dst = (lds_void_t*)(lds_slot + piece * 1024);
    if (piece < CFG::NIH_A)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, dst, 16, voff[p], kbyte, 0, HGEMM_DMA_AUX);
    else
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, dst, 16, voff[p], kbyte, 0, HGEMM_DMA_AUX);
  }
}

// wait until at most `ahead` younger half-tiles (P pieces each) are still in flight
template <class CFG>
__device__ __forceinline__ void pp_wait_ahead(int ahead) {
  if (ahead >= 2)
    wait_vmcnt<2 * CFG::P>();
  else if (ahead == 1)
    wait_vmcnt<CFG::P>();
  else
    wait_vmcnt<0>();
}

__device__ __forceinline__ void pp_barrier() {
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
}
#endif  // __HIP_DEVICE_COMPILE__

template <class CFG, bool SPLITK>
__global__ void __launch_bounds__(CFG::THREADS) hgemm_tn_pp_kernel(const GemmArgs g) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int BM = CFG::BM, BN = CFG::BN, FM = CFG::FM, FN = CFG::FN, NW = CFG::NW, P = CFG::P;
  constexpr int HROW = CFG::HROW_BYTES;

  __shared__ __attribute__((aligned(1024))) char smem[CFG::LDS_BYTES];

  const int tid  = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave_m = wave / CFG::WN;
  const int wave_n = wave % CFG::WN;
  const int grp = wave >> 2;  // 0: leading group, 1: trailing group (one barrier behind)

  const TileCoord tc = map_block(g, BM, BN);
  const int NU = tc.nk * 2;  // half-tiles in this block's K range

  // ---- LDS-DMA source addressing (per-lane, constant over K) -----------------------------------
  const f16* a_base = g.A + (size_t)tc.m0 * g.lda;
  const f16* b_base = g.Bt + (size_t)tc.n0 * g.ldb;
  __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)a_base, 0, 0xFFFFFFFFu, 0x00020000);
  __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)b_base, 0, 0xFFFFFFFFu, 0x00020000);
  uint32_t voff[P];
#pragma unroll
  for (int p = 0; p < P; ++p) {
    const int piece = wave + p * NW;
    const bool isA  = piece < CFG::NIH_A;
    const int il    = isA ? piece : piece - CFG::NIH_A;
    const int r     = il * 16 + (lane >> 2);               // tile row written by this lane
    const int rmax  = isA ? (g.M - 1 - tc.m0) : (g.N - 1 - tc.n0);
    const int rc    = min(r, rmax);
    const int ld    = isA ? g.lda : g.ldb;
    const int chunk = (lane & 3) ^ ((lane >> 5) << 1);     // slot ^ (((r >> 3) & 1) << 1), r & 15 = lane >> 2
    voff[p] = ((uint32_t)rc * (uint32_t)ld + (uint32_t)chunk * 8u) * 2u;
  }

  // ---- fragment read offset inside a ring slot ---------------------------------------------------
  const int l15 = lane & 15;
  const int frag_off = l15 * HROW + (((lane >> 4) ^ ((l15 >> 3) << 1)) << 4);
  const int a_off = wave_m * CFG::TM * HROW + frag_off;
  const int b_off = BM * HROW + wave_n * CFG::TN * HROW + frag_off;

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- prologue: half-tiles 0..2 in flight, wait for half-tile 0 ---------------------------------
  uint32_t kbyte = (uint32_t)tc.k_begin * 2u;   // byte offset of the next half-tile to issue
#pragma unroll
  for (int v = 0; v < 3; ++v) {
    if (v < NU) {
      pp_stage_half<CFG>(rsA, rsB, voff, smem + v * CFG::HALF_BYTES, wave, kbyte);
      kbyte += HROW;
    }
  }
  pp_wait_ahead<CFG>(min(NU - 1, 2));
  pp_barrier();
  if (grp == 1) pp_barrier();  // trailing group starts one barrier later

  for (int u = 0; u < NU; ++u) {
    const char* rd = smem + (u & 3) * CFG::HALF_BYTES;
    char* wr = smem + ((u + 3) & 3) * CFG::HALF_BYTES;
    // R(u)
    f16x8 af[FM], bf[FN];
#pragma unroll
    for (int i = 0; i < FM; ++i) af[i] = *(const f16x8*)(rd + a_off + i * 16 * HROW);
#pragma unroll
    for (int j = 0; j < FN; ++j) bf[j] = *(const f16x8*)(rd + b_off + j * 16 * HROW);
    // trailing group: half-tile u+1 must have landed before the barrier that ends this interval
    if (grp == 1) pp_wait_ahead<CFG>(min(NU - 2 - u, 1));
    pp_barrier();
    // M(u), with the refill of slot (u+3)%4 interleaved
    __builtin_amdgcn_s_setprio(1);
    const bool issue = (u + 3 < NU) && !(g.debug & 1);
    pp_mfma_phase<CFG>(acc, af, bf, rsA, rsB, voff, wr, wave, kbyte, issue);
    if (issue) kbyte += HROW;
    __builtin_amdgcn_s_setprio(0);
    // leading group: same requirement, two younger half-tiles may stay in flight
    if (grp == 0) pp_wait_ahead<CFG>(min(NU - 2 - u, 2));
    pp_barrier();
  }
  if (grp == 0) pp_barrier();  // match the trailing group's extra barrier

  store_tile<16, FM, FN, CFG::TM, CFG::TN, SPLITK>(g, tc, wave_m, wave_n, lane, acc);
#endif  // __HIP_DEVICE_COMPILE__
}

// ------------------------------------------------------------------------------------------------
// MODE 1: "complementary pipelining".  Same half-tile ring, addressing and DMA interleave as above,
// but ONE barrier per half-tile and both wave groups do reads AND MFMAs in every interval, in
// opposite order:
//
//        interval u:   G0:  M(u) ............ R(u+1)          (fragments of u were read at the end
//                      G1:  R(u) ... M(u) ............         of interval u-1: software-pipelined)
//
// After each barrier G0 feeds the matrix pipe at once while G1's LDS reads are in flight; when G0
// runs out of MFMAs and issues its reads for the next half-tile, G1 is mid-stream.  The pipe
// always has a wave with work, and no second register set is needed (G0 reuses the fragment
// registers as the MFMAs retire them).
//   visibility: at the barrier that ends interval u every wave has waited until at most half-tile
//   u+3 (issued during interval u) is in flight, so half-tiles <= u+2 are visible in interval u+1
//   (G0 reads u+2 at its end, G1 reads u+1 at its start).
//   refill: slot (u+3)%4 held half-tile u-1, last read by G1 at the start of interval u-1 and
//   retired before its MFMAs there; the refill is issued in interval u, one barrier later.
template <class CFG, bool SPLITK>
__global__ void __launch_bounds__(CFG::THREADS) hgemm_tn_cp_kernel(const GemmArgs g) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int BM = CFG::BM, BN = CFG::BN, FM = CFG::FM, FN = CFG::FN, NW = CFG::NW, P = CFG::P;
  constexpr int HROW = CFG::HROW_BYTES;

  __shared__ __attribute__((aligned(1024))) char smem[CFG::LDS_BYTES];

  const int tid  = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave_m = wave / CFG::WN;
  const int wave_n = wave % CFG::WN;
  const int grp = wave >> 2;  // 0: software-pipelined group, 1: read-then-compute group

  const TileCoord tc = map_block(g, BM, BN);
  const int NU = tc.nk * 2;

  const f16* a_base = g.A + (size_t)tc.m0 * g.lda;
  const f16* b_base = g.Bt + (size_t)tc.n0 * g.ldb;
  __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)a_base, 0, 0xFFFFFFFFu, 0x00020000);
  __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)b_base, 0, 0xFFFFFFFFu, 0x00020000);
  uint32_t voff[P];
#pragma unroll
  for (int p = 0; p < P; ++p) {
    const int piece = wave + p * NW;
    const bool isA  = piece < CFG::NIH_A;
    const int il    = isA ? piece : piece - CFG::NIH_A;
    const int r     = il * 16 + (lane >> 2);
    const int rmax  = isA ? (g.M - 1 - tc.m0) : (g.N - 1 - tc.n0);
    const int rc    = min(r, rmax);
    const int ld    = isA ? g.lda : g.ldb;
    const int chunk = (lane & 3) ^ ((lane >> 5) << 1);
    voff[p] = ((uint32_t)rc * (uint32_t)ld + (uint32_t)chunk * 8u) * 2u;
  }

  const int l15 = lane & 15;
  const int frag_off = l15 * HROW + (((lane >> 4) ^ ((l15 >> 3) << 1)) << 4);
  const int a_off = wave_m * CFG::TM * HROW + frag_off;
  const int b_off = BM * HROW + wave_n * CFG::TN * HROW + frag_off;

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // prologue: half-tiles 0..2 in flight; 0 and 1 must be visible in interval 0
  uint32_t kbyte = (uint32_t)tc.k_begin * 2u;
#pragma unroll
  for (int v = 0; v < 3; ++v) {
    if (v < NU) {
      pp_stage_half<CFG>(rsA, rsB, voff, smem + v * CFG::HALF_BYTES, wave, kbyte);
      kbyte += HROW;
    }
  }
  pp_wait_ahead<CFG>(NU >= 3 ? 1 : 0);
  pp_barrier();

  f16x8 af[FM], bf[FN];
  if (grp == 0) {  // G0 enters the loop with the fragments of half-tile 0 in registers
#pragma unroll
    for (int i = 0; i < FM; ++i) af[i] = *(const f16x8*)(smem + a_off + i * 16 * HROW);
#pragma unroll
    for (int j = 0; j < FN; ++j) bf[j] = *(const f16x8*)(smem + b_off + j * 16 * HROW);
  }

  for (int u = 0; u < NU; ++u) {
    char* wr = smem + ((u + 3) & 3) * CFG::HALF_BYTES;
    const bool issue = (u + 3 < NU) && !(g.debug & 1);
    if (grp == 1) {  // R(u) first
      const char* rd = smem + (u & 3) * CFG::HALF_BYTES;
#pragma unroll
      for (int i = 0; i < FM; ++i) af[i] = *(const f16x8*)(rd + a_off + i * 16 * HROW);
#pragma unroll
      for (int j = 0; j < FN; ++j) bf[j] = *(const f16x8*)(rd + b_off + j * 16 * HROW);
    }
    __builtin_amdgcn_sched_barrier(0);
    pp_mfma_phase<CFG>(acc, af, bf, rsA, rsB, voff, wr, wave, kbyte, issue);
    if (issue) kbyte += HROW;
    __builtin_amdgcn_sched_barrier(0);
    if (grp == 0 && u + 1 < NU) {  // R(u+1) behind the MFMAs
      const char* rd = smem + ((u + 1) & 3) * CFG::HALF_BYTES;
#pragma unroll
      for (int i = 0; i < FM; ++i) af[i] = *(const f16x8*)(rd + a_off + i * 16 * HROW);
#pragma unroll
      for (int j = 0; j < FN; ++j) bf[j] = *(const f16x8*)(rd + b_off + j * 16 * HROW);
    }
    // everything up to half-tile u+2 must have landed before the next interval
    pp_wait_ahead<CFG>(issue ? 1 : 0);
    pp_barrier();
  }

  store_tile<16, FM, FN, CFG::TM, CFG::TN, SPLITK>(g, tc, wave_m, wave_n, lane, acc);
#endif  // __HIP_DEVICE_COMPILE__
}

}  // namespace hgemm_mi355x
