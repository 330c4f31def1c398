// tools/xcd_probe.hip -- which XCD does block b of a 1-D grid run on?  (crf_batch_frame_kernel assumes b % 8, for speed only.)
// hipcc --offload-arch=gfx950 -O2 tools/xcd_probe.hip -o /tmp/xcd_probe && /tmp/xcd_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(int *out, size_t lds_dummy) {
    extern __shared__ char lds[];
    if (threadIdx.x == 0) {
        unsigned x;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
        out[blockIdx.x] = (int)(x & 0xf);
        lds[0] = (char)lds_dummy;
    }
}
int main() {
    for (int nblk : {64, 1024, 4096}) {
        int *d; hipMalloc(&d, nblk * sizeof(int));
        hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 40960);
        hipLaunchKernelGGL(probe, dim3(nblk), dim3(256), 40960, 0, d, (size_t)0);
        std::vector<int> h(nblk);
        hipMemcpy(h.data(), d, nblk * sizeof(int), hipMemcpyDeviceToHost);
        int match = 0;
        for (int b = 0; b < nblk; ++b) match += h[b] == b % 8;
        printf("grid %d: %d of %d blocks on XCD b %% 8; first 24:", nblk, match, nblk);
        for (int b = 0; b < 24; ++b) printf(" %d", h[b]);
        printf("\n");
        hipFree(d);
    }
    return 0;
}
