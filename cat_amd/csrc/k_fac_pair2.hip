// cat_amd/csrc/k_fac_pair2.hip -- factored register-resident recursions: two utterances per workgroup (768-thread geometries and the 512 x 30 layout of its own)
// (explicit instantiations of what the host side launches; definitions in k_fac_body.h)
#include "k_fac_body.h"

namespace crf {

#define CRF_INST_FAC(FLAG)                                                                                                                  \
    template __global__ void crf_fac_pair2_kernel<FLAG, kFac3Threads, kFac3ArcCh, CRF_FAC3_NB2, CRF_FAC3_NB2, false, true>(FacParams, FacParams);      \
    template __global__ void crf_fac_pair2_kernel<FLAG, kFac3Threads, kFac3ArcCh, CRF_FAC3_NB2, CRF_FAC3_NB2, true, true>(FacParams, FacParams);       \
    template __global__ void crf_fac_pair2_kernel<FLAG, kFac3Threads, kFac3NCH, CRF_FAC3_NB2, CRF_FAC3_NB2, false, false>(FacParams, FacParams);       \
    template __global__ void crf_fac_pair2_kernel<FLAG, kFac3Threads, kFac3NCH, CRF_FAC3_NB2, CRF_FAC3_NB2, true, false>(FacParams, FacParams);        \
    template __global__ void crf_fac_pair2_kernel<FLAG, kResThreads, kResNCH, CRF_FAC5_NB2, CRF_FAC5_NB2, false, true>(FacParams, FacParams);          \
    template __global__ void crf_fac_pair2_kernel<FLAG, kResThreads, kResNCH, CRF_FAC5_NB2, CRF_FAC5_NB2, true, true>(FacParams, FacParams);
CRF_INST_FAC(true)
CRF_INST_FAC(false)
#undef CRF_INST_FAC
#if CRF_FAC3L_NCH != 20
#define CRF_INST_FAC21(FLAG)                                                                                                                \
    template __global__ void crf_fac_pair2_kernel<FLAG, kFac3Threads, kFac3LNCH, CRF_FAC3_NB2, CRF_FAC3_NB2, false, true>(FacParams, FacParams);       \
    template __global__ void crf_fac_pair2_kernel<FLAG, kFac3Threads, kFac3LNCH, CRF_FAC3_NB2, CRF_FAC3_NB2, true, true>(FacParams, FacParams);
CRF_INST_FAC21(true)
CRF_INST_FAC21(false)
#undef CRF_INST_FAC21
#endif

}  // namespace crf
