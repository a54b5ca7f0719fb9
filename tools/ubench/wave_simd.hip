// Which SIMD does each wave of a 512-thread workgroup land on?  (HW_REG_HW_ID.simd_id)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(512) k(int* out) {
    extern __shared__ float lds[];
    const int wave = threadIdx.x >> 6;
    unsigned hw = __builtin_amdgcn_s_getreg((15 << 11) | (0 << 6) | 4);   // HW_ID bits [15:0]
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + wave] = hw;
}
int main() {
    int* d; hipMalloc(&d, 256 * 8 * 4);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 161216, 0, d);
        hipDeviceSynchronize();
        int h[256 * 8]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        int hist[5] = {0};
        for (int b = 0; b < 256; ++b) {
            int cnt[4] = {0};
            for (int w = 0; w < 8; ++w) cnt[(h[b * 8 + w] >> 4) & 3]++;
            int mx = 0; for (int i = 0; i < 4; ++i) mx = cnt[i] > mx ? cnt[i] : mx;
            hist[mx > 4 ? 4 : mx]++;
            if (b < 6) { printf("block %d: simd of waves 0..7:", b); for (int w = 0; w < 8; ++w) printf(" %d", (h[b * 8 + w] >> 4) & 3); printf("\n"); }
        }
        printf("max waves on one SIMD -> #blocks: 2:%d 3:%d 4+:%d\n", hist[2], hist[3], hist[4]);
    }
    return 0;
}
