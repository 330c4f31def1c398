// cat_amd/csrc/k_fac_768.hip -- factored register-resident recursions: the 768-thread geometries (row constants in registers or in the LDS table, 20 / 21 chunks), the 512-thread fallback, two CUs per recursion
// (explicit instantiations of what the host side launches; definitions in k_fac_body.h)
#include "k_fac_body.h"

namespace crf {

#define CRF_INST_FAC(FLAG)                                                                                                                  \
    template __global__ void crf_fac_pair_kernel<FLAG, kFac3Threads, kFac3ArcCh, CRF_FAC3_NB_F, CRF_FAC3_NB_B, false, true>(FacParams, FacParams);     \
    template __global__ void crf_fac_pair_kernel<FLAG, kFac3Threads, kFac3ArcCh, CRF_FAC3_NB_F, CRF_FAC3_NB_B, true, true>(FacParams, FacParams);      \
    template __global__ void crf_fac_pair_kernel<FLAG, kFac3Threads, kFac3NCH, CRF_FAC3_NB_F, CRF_FAC3_NB_B, false, false>(FacParams, FacParams);      \
    template __global__ void crf_fac_pair_kernel<FLAG, kFac3Threads, kFac3NCH, CRF_FAC3_NB_F, CRF_FAC3_NB_B, true, false>(FacParams, FacParams);       \
    template __global__ void crf_fac_pair_kernel<FLAG, kResThreads, kResNCH, 6, 6, true, false>(FacParams, FacParams);
CRF_INST_FAC(true)
CRF_INST_FAC(false)
#undef CRF_INST_FAC
#if CRF_FAC3L_NCH != 20   // (table geometry with all 21 chunk slots holding arcs: instantiations of its own unless an A/B build makes it the 20-chunk one)
#define CRF_INST_FAC21(FLAG)                                                                                                                \
    template __global__ void crf_fac_pair_kernel<FLAG, kFac3Threads, kFac3LNCH, CRF_FAC3_NB_F, CRF_FAC3_NB_B, false, true>(FacParams, FacParams);      \
    template __global__ void crf_fac_pair_kernel<FLAG, kFac3Threads, kFac3LNCH, CRF_FAC3_NB_F, CRF_FAC3_NB_B, true, true>(FacParams, FacParams);
CRF_INST_FAC21(true)
CRF_INST_FAC21(false)
#undef CRF_INST_FAC21
#endif
template __global__ void crf_fac2_pair_kernel<kFac3Threads, kFac3ArcCh, CRF_FAC3_NB_F, CRF_FAC3_NB_B>(FacParams, FacParams);

}  // namespace crf
