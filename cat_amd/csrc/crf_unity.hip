// cat_amd/csrc/crf_unity.hip -- every kernel family and the host side as ONE translation unit: timing builds (-DCRF_TIMING: the stamp buffer
// g_tm is one device global, crf_device.h), and a way to check that the families do not depend on the order they are compiled in.
#include "k_chain.hip"
#include "k_res.hip"
#include "k_fac_1024.hip"
#include "k_fac_768.hip"
#include "k_fac_pair2.hip"
#include "k_grad.hip"
#include "k_batch.hip"
#include "k_robust.hip"
#include "crf_host.hip"
