// oracle/ref_mgpu_stub.h -- TEST INFRASTRUCTURE ONLY.
// The reference's numerator kernels (/root/reference/src/ctc_crf/gpu_ctc/gpu_ctc_kernels.h:87-458) include two headers of the vendored
// moderngpu library (gpu_ctc_kernels.h:3-4) and use exactly four of its entities:
//   mgpu::CTAScan<NT>::{Storage, Scan}     gpu_ctc_kernels.h:16, 43     exclusive add-scan over the NT threads of a block, total returned
//   mgpu::CTAMergesort<NT, VT, true, true, int, int, less>   :294-295   stable sort of the block's NT x VT (key, value) pairs (blocked order:
//                                                                        thread tid holds elements [VT tid, VT tid + VT)); afterwards the sorted
//                                                                        keys lie in keys_shared and each thread holds its sorted values
//   mgpu::less<int>                        :294                         a < b
//   popc                                   :43                          population count
// moderngpu itself cannot be built for gfx950 (inline PTX, WARP_SIZE = 32: contrib/moderngpu/include/device/intrinsics.cuh:86-101,
// devicetypes.cuh:56), so oracle/Makefile target `ref` redirects those two includes to THIS file (clang -ivfsoverlay) and compiles the
// reference's own ctc_entrypoint.cu / gpu_ctc.h / gpu_ctc_kernels.h / ctc_helper.h unmodified, in place, for wave64.  Own code, written from the
// interface the call sites need; what the library computes there (a label sort and a scan) affects the order of summation only.
#pragma once
#include <hip/hip_runtime.h>
#include <climits>

namespace mgpu {

template <typename T>
struct less {
    __host__ __device__ bool operator()(T a, T b) const { return a < b; }
};

__host__ __device__ inline int popc(unsigned x) { return __builtin_popcount(x); }

// exclusive add-scan of one int per thread over the NT threads of the block; *total = the sum (the call site passes shared memory)
template <int NT>
struct CTAScan {
    struct Storage { int shared[2 * NT]; };
    __device__ static int Scan(int tid, int x, Storage &storage, int *total) {
        storage.shared[tid] = x;
        int first = 0;
        __syncthreads();
        for (int offset = 1; offset < NT; offset += offset) {   // Hillis-Steele, two buffers in rotation
            if (tid >= offset) x += storage.shared[first + tid - offset];
            first = NT - first;
            storage.shared[first + tid] = x;
            __syncthreads();
        }
        *total = storage.shared[first + NT - 1];
        const int excl = tid ? storage.shared[first + tid - 1] : 0;
        __syncthreads();
        return excl;
    }
};

// stable sort of the block's NT * VT pairs by rank (NV <= 1280 here: a quadratic count per element is nothing for a checker).  Elements at and
// beyond `count` carry INT_MAX keys at the call site (gpu_ctc_kernels.h:268-270), so sorting all NV elements leaves the first `count` exactly
// where a sort of only those would.
template <int NT, int VT, bool Stable, bool HasValues, typename KeyType, typename ValType, typename Comp>
__device__ void CTAMergesort(KeyType threadKeys[VT], ValType threadValues[VT], KeyType *keys_shared, ValType *values_shared, int count, int tid, Comp comp) {
    (void)count;
    constexpr int NV = NT * VT;
    for (int i = 0; i < VT; ++i) { keys_shared[VT * tid + i] = threadKeys[i]; if (HasValues) values_shared[VT * tid + i] = threadValues[i]; }
    __syncthreads();
    int rank[VT];
    for (int i = 0; i < VT; ++i) {
        const int me = VT * tid + i;
        const KeyType k = threadKeys[i];
        int r = 0;
        for (int j = 0; j < NV; ++j) {
            const KeyType o = keys_shared[j];
            r += (comp(o, k) || (!comp(k, o) && j < me)) ? 1 : 0;   // strictly smaller, or equal and earlier: stable
        }
        rank[i] = r;
    }
    __syncthreads();
    for (int i = 0; i < VT; ++i) { keys_shared[rank[i]] = threadKeys[i]; if (HasValues) values_shared[rank[i]] = threadValues[i]; }
    __syncthreads();
    for (int i = 0; i < VT; ++i) { threadKeys[i] = keys_shared[VT * tid + i]; if (HasValues) threadValues[i] = values_shared[VT * tid + i]; }
    __syncthreads();
}

}  // namespace mgpu
