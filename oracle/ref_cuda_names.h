// oracle/ref_cuda_names.h -- TEST INFRASTRUCTURE ONLY.
// Force-included when hipcc compiles the REFERENCE file /root/reference/src/ctc_crf/gpu_den/den_calculate.cu
// in place (oracle/Makefile target `ref`).  Maps exactly the CUDA-runtime names that file uses
// (den_calculate.cu:16-25, 375-390, 394-410, 427-481) onto the HIP runtime so the reference's own
// kernels can serve as a GPU-side oracle.  Never included by anything under cat_amd/.
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
