// LDS read bandwidth of one CU for ds_read_b128 / ds_read_b64 with 1, 2, 4, 8 waves issuing (conflict-free, linear
// addresses): cycles per instruction per wave and bytes per cycle per CU.  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int W>
__global__ void k_rd(unsigned long long* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) ((float*)lds)[i] = (float)i;
    __syncthreads();
    const char* p = lds + 4096 * wave + (W == 16 ? 16 : 8) * lane;
    f32x4 acc = {0, 0, 0, 0};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (W == 16) { f32x4 v; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"((unsigned)(size_t)p), "i"(0)); asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory"); acc += v; }
            else { f32x2 v; asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"((unsigned)(size_t)p)); asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory"); acc.x += v.x; }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) { out[2 * wave] = t1 - t0; out[2 * wave + 1] = (unsigned long long)acc.x; }
}
int main() {
    unsigned long long* d; hipMalloc(&d, 256);
    const int iters = 512;
    for (int w : {16, 8})
        for (int waves : {1, 2, 4, 8}) {
            if (w == 16) hipLaunchKernelGGL(k_rd<16>, dim3(1), dim3(64 * waves), 65536, 0, d, iters);
            else hipLaunchKernelGGL(k_rd<8>, dim3(1), dim3(64 * waves), 65536, 0, d, iters);
            unsigned long long r[16]; hipMemcpy(r, d, 128, hipMemcpyDeviceToHost);
            double worst = 0; for (int i = 0; i < waves; ++i) worst = r[2 * i] > worst ? r[2 * i] : worst;
            const double cpi = worst / (iters * 8.0);
            printf("ds_read_b%d, %d waves: %.1f cycles per instruction per wave, %.0f B/clk per CU\n", 8 * w, waves, cpi,
                   waves * 64.0 * w / cpi);
        }
    return 0;
}
