// oracle/ref_cuda_names.h -- TEST INFRASTRUCTURE ONLY.
// Force-included when hipcc compiles the REFERENCE files /root/reference/src/ctc_crf/gpu_den/den_calculate.cu and
// gpu_ctc/ctc_entrypoint.cu in place (oracle/Makefile target `ref`).  Maps exactly the CUDA-runtime names those files use
// (den_calculate.cu:16-25, 375-390, 394-410, 427-481; gpu_ctc.h:12, 149, 180-228, 272-277, 364-369; ctc.h:14) onto the HIP runtime so the
// reference's own kernels can serve as a GPU-side oracle.  Never included by anything under cat_amd/.
#pragma once
#include <hip/hip_runtime.h>
#include <cstring>
#include <cmath>
#define cudaError_t hipError_t
#define cudaSuccess hipSuccess
#define cudaGetErrorString hipGetErrorString
#define cudaGetDevice hipGetDevice
#define cudaSetDevice hipSetDevice
#define cudaMalloc hipMalloc
#define cudaFree hipFree
#define cudaMemcpy hipMemcpy
#define cudaMemcpyHostToDevice hipMemcpyHostToDevice
#define cudaStream_t hipStream_t
#define cudaMemcpyAsync hipMemcpyAsync
#define cudaMemcpyDeviceToHost hipMemcpyDeviceToHost
#define cudaStreamSynchronize hipStreamSynchronize
#define cudaGetLastError hipGetLastError
#define CUstream_st ihipStream_t   /* ctc.h:14 `typedef struct CUstream_st* CUstream;` becomes hipStream_t */
