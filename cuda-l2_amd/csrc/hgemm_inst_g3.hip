// Kernel instantiations, group 3 of hgemm_configs.def (split so the groups build in parallel).
#include "hgemm_launch.hpp"

namespace hgemm_mi355x {
#define HGEMM_INST_0(...)
#define HGEMM_INST_1(...)
#define HGEMM_INST_2(...)
#define HGEMM_INST_3(...)
#undef HGEMM_INST_3
#define HGEMM_INST_3(BM, BN, WM, WN, MI, NB) \
  template void launch_cfg<Cfg<BM, BN, WM, WN, MI, NB>>(const GemmArgs&, int, hipStream_t, int, TimingSlot);
#define HGEMM_CFG(G, BM, BN, WM, WN, MI, NB) HGEMM_INST_##G(BM, BN, WM, WN, MI, NB)
#define HGEMM_SPINST_0(...)
#define HGEMM_SPINST_1(...)
#define HGEMM_SPINST_2(...)
#define HGEMM_SPINST_3(...)
#undef HGEMM_SPINST_3
#define HGEMM_SPINST_3(BM, BN, WM, WN, MI) \
  template void launch_sp<CfgSP<BM, BN, WM, WN, MI>>(const GemmArgs&, int, hipStream_t, int, TimingSlot);
#define HGEMM_SP(G, BM, BN, WM, WN, MI) HGEMM_SPINST_##G(BM, BN, WM, WN, MI)
#define HGEMM_SQINST_0(...)
#define HGEMM_SQINST_1(...)
#define HGEMM_SQINST_2(...)
#define HGEMM_SQINST_3(...)
#undef HGEMM_SQINST_3
#define HGEMM_SQINST_3(BM, BN, WM, WN, KT, MI) \
  template void launch_sq<CfgSQ<BM, BN, WM, WN, KT, MI>>(const GemmArgs&, int, hipStream_t, int, TimingSlot);
#define HGEMM_SQ(G, BM, BN, WM, WN, KT, MI) HGEMM_SQINST_##G(BM, BN, WM, WN, KT, MI)
#define HGEMM_RSINST_0(...)
#define HGEMM_RSINST_1(...)
#define HGEMM_RSINST_2(...)
#define HGEMM_RSINST_3(...)
#undef HGEMM_RSINST_3
#define HGEMM_RSINST_3(BM, BN, BKS, LB) \
  template void launch_rs<CfgRS<BM, BN, BKS, LB>>(const GemmArgs&, int, hipStream_t, int, TimingSlot);
#define HGEMM_RS(G, BM, BN, BKS, LB) HGEMM_RSINST_##G(BM, BN, BKS, LB)
#define HGEMM_WDINST_0(...)
#define HGEMM_WDINST_1(...)
#define HGEMM_WDINST_2(...)
#define HGEMM_WDINST_3(...)
#undef HGEMM_WDINST_3
#define HGEMM_WDINST_3(FM, FN, KW) \
  template void launch_wd<CfgWD<FM, FN, KW>>(const GemmArgs&, int, hipStream_t, int, TimingSlot);
#define HGEMM_WD(G, FM, FN, KW) HGEMM_WDINST_##G(FM, FN, KW)
#include "hgemm_configs.def"
#undef HGEMM_CFG
#undef HGEMM_SP
#undef HGEMM_SQ
#undef HGEMM_RS
#undef HGEMM_WD
}  // namespace hgemm_mi355x
