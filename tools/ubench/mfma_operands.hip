// Microbenchmark: issue rate of v_mfma_f32_16x16x4_f32 as a function of where its A/B
// operands and accumulators live (same register, same VGPR bank, different banks, AGPR acc).
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_operands mfma_operands.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>

#define BODY(NAME, I0, I1, I2, I3, I4, I5, I6, I7)                                              \
    __global__ void __launch_bounds__(512) NAME(float* out, int iters) {                        \
        asm volatile(                                                                           \
            "v_mov_b32 v20, 1.0\n v_mov_b32 v21, 2.0\n v_mov_b32 v22, 0.5\n v_mov_b32 v23, 0.25\n" \
            "v_mov_b32 v24, 1.0\n v_mov_b32 v25, 2.0\n v_mov_b32 v26, 0.5\n v_mov_b32 v27, 0.25\n" \
            "v_mov_b32 v28, 1.0\n v_mov_b32 v29, 2.0\n v_mov_b32 v30, 0.5\n v_mov_b32 v31, 0.25\n" \
            "v_mov_b32 v0, 0\n v_mov_b32 v1, 0\n v_mov_b32 v2, 0\n v_mov_b32 v3, 0\n"            \
            "v_mov_b32 v4, 0\n v_mov_b32 v5, 0\n v_mov_b32 v6, 0\n v_mov_b32 v7, 0\n"            \
            "v_accvgpr_write_b32 a0, 0\n v_accvgpr_write_b32 a1, 0\n v_accvgpr_write_b32 a2, 0\n v_accvgpr_write_b32 a3, 0\n" \
            "v_accvgpr_write_b32 a4, 0\n v_accvgpr_write_b32 a5, 0\n v_accvgpr_write_b32 a6, 0\n v_accvgpr_write_b32 a7, 0\n" \
            "s_nop 4\n"                                                                         \
            "1:\n" I0 "\n" I1 "\n" I2 "\n" I3 "\n" I4 "\n" I5 "\n" I6 "\n" I7 "\n"              \
            "s_sub_u32 %1, %1, 1\n s_cmp_lg_u32 %1, 0\n s_cbranch_scc1 1b\n"                    \
            "s_nop 7\n s_nop 7\n"                                                               \
            "v_accvgpr_read_b32 v8, a0\n s_nop 2\n v_add_f32 v0, v0, v8\n"                       \
            "v_add_f32 v0, v0, v4\n global_store_dword %0, v0, off\n s_waitcnt vmcnt(0)\n"       \
            : : "v"(out + blockIdx.x * 512 + threadIdx.x), "s"(iters)                            \
            : "v0","v1","v2","v3","v4","v5","v6","v7","v8","v20","v21","v22","v23","v24","v25","v26","v27", \
              "v28","v29","v30","v31","a0","a1","a2","a3","a4","a5","a6","a7","scc","memory");   \
    }

#define M(acc, a, b) "v_mfma_f32_16x16x4_f32 " acc ", " a ", " b ", " acc
// 1. A == B register
BODY(k_same, M("v[0:3]","v20","v20"), M("v[4:7]","v20","v20"), M("v[0:3]","v21","v21"), M("v[4:7]","v21","v21"),
     M("v[0:3]","v22","v22"), M("v[4:7]","v22","v22"), M("v[0:3]","v23","v23"), M("v[4:7]","v23","v23"))
// 2. A, B different registers in the SAME bank (index mod 4 equal)
BODY(k_samebank, M("v[0:3]","v20","v24"), M("v[4:7]","v20","v24"), M("v[0:3]","v21","v25"), M("v[4:7]","v21","v25"),
     M("v[0:3]","v22","v26"), M("v[4:7]","v22","v26"), M("v[0:3]","v23","v27"), M("v[4:7]","v23","v27"))
// 3. A, B in different banks
BODY(k_diffbank, M("v[0:3]","v20","v25"), M("v[4:7]","v20","v25"), M("v[0:3]","v21","v26"), M("v[4:7]","v21","v26"),
     M("v[0:3]","v22","v27"), M("v[4:7]","v22","v27"), M("v[0:3]","v23","v24"), M("v[4:7]","v23","v24"))
// 4. like the kernel: each tile its own A register, B shared (pattern a0[c], b[c] / a1[c], b[c])
BODY(k_kernel_like, M("v[0:3]","v20","v28"), M("v[4:7]","v24","v28"), M("v[0:3]","v21","v29"), M("v[4:7]","v25","v29"),
     M("v[0:3]","v22","v30"), M("v[4:7]","v26","v30"), M("v[0:3]","v23","v31"), M("v[4:7]","v27","v31"))
// 5. AGPR accumulators, operands as in 4
BODY(k_agpr_acc, M("a[0:3]","v20","v28"), M("a[4:7]","v24","v28"), M("a[0:3]","v21","v29"), M("a[4:7]","v25","v29"),
     M("a[0:3]","v22","v30"), M("a[4:7]","v26","v30"), M("a[0:3]","v23","v31"), M("a[4:7]","v27","v31"))
// 6. AGPR accumulators and A == B
BODY(k_agpr_same, M("a[0:3]","v20","v20"), M("a[4:7]","v20","v20"), M("a[0:3]","v21","v21"), M("a[4:7]","v21","v21"),
     M("a[0:3]","v22","v22"), M("a[4:7]","v22","v22"), M("a[0:3]","v23","v23"), M("a[4:7]","v23","v23"))

template <class K>
static void run(const char* name, K k, int threads, float* d) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, d, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double mfma_per_simd = (double)iters * 8 * (threads / 256);   // waves per SIMD = threads/256
    const double ns_per = ms * 1e6 / mfma_per_simd;
    printf("%-14s %4d thr: %8.3f ms, %6.2f ns per MFMA per SIMD (= %5.1f cycles @2.4GHz), %6.1f TFLOP/s\n", name,
           threads, ms, ns_per, ns_per * 2.4, 256.0 * 4 * mfma_per_simd * 2048 / (ms * 1e-3) / 1e12);
}

int main() {
    float* d;
    hipMalloc(&d, 256 * 512 * 4);
    for (int t : {256, 512}) {
        run("same", k_same, t, d);
        run("samebank", k_samebank, t, d);
        run("diffbank", k_diffbank, t, d);
        run("kernel_like", k_kernel_like, t, d);
        run("agpr_acc", k_agpr_acc, t, d);
        run("agpr_same", k_agpr_same, t, d);
    }
    return 0;
}
