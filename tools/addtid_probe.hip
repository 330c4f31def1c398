// tools/addtid_probe.hip -- does ds_write_addtid_b32 take its base from ALL of M0 (the 160 KB LDS) or from 16 bits?  hipcc --offload-arch=gfx950 -O2 tools/addtid_probe.hip -o tools/addtid_probe.bin
// (MI355X: bases 69 632 and 100 000 land where they should; beyond the allocation nothing is written: profiles/round6_ab_addtid.txt)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float *out, unsigned m0v) {
    extern __shared__ float x[];
    for (int i = threadIdx.x; i < 40000; i += 64) x[i] = -1.f;
    __syncthreads();
    const float v = 1000.f + threadIdx.x;
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tds_write_addtid_b32 %0 offset:16" : : "v"(v), "s"(m0v) : "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 40000; i += 64) out[i] = x[i];
}
int main() {
    float *d; hipMalloc(&d, 160000);
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160000);
    float *h = (float *)malloc(160000);
    for (unsigned m0v : {0u, 4096u, 65536u + 4096u, 100000u, 0x30000u | 4096u}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 160000, 0, d, m0v);
        hipMemcpy(h, d, 160000, hipMemcpyDeviceToHost);
        int first = -1, n = 0;
        for (int i = 0; i < 40000; ++i) if (h[i] >= 0.f) { if (first < 0) first = i; ++n; }
        printf("m0 = %u: %d floats written, first at float index %d (byte %d) value %.0f\n", m0v, n, first, first * 4, first >= 0 ? h[first] : -1.f);
    }
    return 0;
}
