// cat_amd/csrc/k_fac_1024.hip -- factored register-resident recursions: 1024 threads x 15 chunks, four waves per SIMD (the planner's first choice; the metric graph's kernel)
// (explicit instantiations of what the host side launches; definitions in k_fac_body.h)
#include "k_fac_body.h"

namespace crf {

#define CRF_INST_FAC(FLAG)                                                                                                                  \
    template __global__ void crf_fac_pair_kernel<FLAG, kFac4Threads, kFac4NCH, CRF_FAC4_NB, CRF_FAC4_NB, false, true>(FacParams, FacParams);           \
    template __global__ void crf_fac_pair_kernel<FLAG, kFac4Threads, kFac4NCH, CRF_FAC4_NB_ML, CRF_FAC4_NB_ML, true, true>(FacParams, FacParams);
CRF_INST_FAC(true)
CRF_INST_FAC(false)
#undef CRF_INST_FAC
template __global__ void crf_fac2_pair_kernel<kFac4Threads, kFac4NCH, CRF_FAC4_NB_ML, CRF_FAC4_NB_ML>(FacParams, FacParams);

}  // namespace crf
