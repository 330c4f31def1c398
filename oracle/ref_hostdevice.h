// oracle/ref_hostdevice.h -- TEST INFRASTRUCTURE ONLY.  Stands in for /root/reference/src/ctc_crf/gpu_ctc/hostdevice.h when the reference's
// numerator is compiled by hipcc (oracle/Makefile target `ref`): that header keys HOSTDEVICE on __CUDACC__, which hipcc does not define.
#pragma once
#define HOSTDEVICE __host__ __device__
