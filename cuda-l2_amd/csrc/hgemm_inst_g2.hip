// Kernel instantiations, group 2 of hgemm_configs.def (split so the groups build in parallel).
#include "hgemm_launch.hpp"

namespace hgemm_mi355x {
#define HGEMM_INST_0(...)
#define HGEMM_INST_1(...)
#define HGEMM_INST_2(...)
#define HGEMM_INST_3(...)
#undef HGEMM_INST_2
#define HGEMM_INST_2(BM, BN, WM, WN, MI, NB) \
  template void launch_cfg<Cfg<BM, BN, WM, WN, MI, NB>>(const GemmArgs&, int, hipStream_t, bool);
#define HGEMM_CFG(G, BM, BN, WM, WN, MI, NB) HGEMM_INST_##G(BM, BN, WM, WN, MI, NB)
#include "hgemm_configs.def"
#undef HGEMM_CFG
}  // namespace hgemm_mi355x
