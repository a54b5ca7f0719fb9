// Microbenchmark: MFMA issue rate of v_mfma_f32_16x16x4_f32 when every block of 8 MFMAs is
// fed by LDS reads (the fused kernel's inner-loop shape), 1 or 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int PATTERN, bool DEP>
__global__ void __launch_bounds__(512) k(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 32768; i += blockDim.x) lds[i] = 0.001f * (i & 255);
    __syncthreads();
    const int s = lane & 15, q = lane >> 4;
    const float* xb = lds + s * 136 + 4 * q;                       // B rows, stride 136
    const float* wa0 = lds + 8192 + (16 * wave + s) * 132 + 4 * q;  // A rows, stride 132
    const float* wa1 = wa0 + 64 * 132;
    const float* wc0 = lds + 8192 + (4 * q) * 132 + 16 * (wave & 3) + s;   // A columns
    const float* wc1 = wc0 + 64;
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = acc0;
    f32x4 cb = {1.f, 2.f, 3.f, 4.f}, ca0 = {0.5f, 0.25f, 2.f, 1.f}, ca1 = ca0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            f32x4 b, a0, a1;
            if (PATTERN == 0) { b = cb; a0 = ca0; a1 = ca1; }
            if (PATTERN == 1) {            // forward: 3 x ds_read_b128
                b = *(const f32x4*)(xb + 16 * u);
                a0 = *(const f32x4*)(wa0 + 16 * u);
                a1 = *(const f32x4*)(wa1 + 16 * u);
            }
            if (PATTERN == 2) {            // reverse: 1 x b128 + 8 x b32 (compiler pairs them)
                b = *(const f32x4*)(xb + 16 * u);
                const float* p0 = wc0 + 16 * u * 132;
                const float* p1 = wc1 + 16 * u * 132;
                a0 = f32x4{p0[0], p0[132], p0[264], p0[396]};
                a1 = f32x4{p1[0], p1[132], p1[264], p1[396]};
            }
            if (!DEP) { asm volatile("" ::"v"(b), "v"(a0), "v"(a1)); b = cb; a0 = ca0; a1 = ca1; }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[c], b[c], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[c], b[c], acc1, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    out[blockIdx.x * 512 + threadIdx.x] = acc0.x + acc1.y;
}

template <class K>
static void run(const char* name, K kk, int threads, float* d) {
    const int iters = 2000;
    hipFuncSetAttribute((const void*)kk, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kk, dim3(256), dim3(threads), 140000, 0, d, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kk, dim3(256), dim3(threads), 140000, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double mf = (double)iters * 64 * (threads / 256);
    printf("%-28s %4d thr: %7.3f ms, %6.1f cycles@2.4GHz per MFMA per SIMD (ideal 32)\n", name, threads, ms,
           ms * 1e6 / mf * 2.4);
}
int main() {
    float* d; hipMalloc(&d, 256 * 512 * 4);
    for (int t : {256, 512}) {
        run("const operands", k<0, true>, t, d);
        run("fwd loads, MFMA independent", k<1, false>, t, d);
        run("fwd loads, MFMA dependent", k<1, true>, t, d);
        run("rev loads, MFMA independent", k<2, false>, t, d);
        run("rev loads, MFMA dependent", k<2, true>, t, d);
    }
    return 0;
}
