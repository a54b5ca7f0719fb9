// Where does v_mfma_f32_4x4x1_16B_f32 take its B operand from, and does it matter?  k_solve_bcast (cnf_bcast.hip) keeps 240 resident
// B registers per lane in the ACCUMULATOR half of the register file and lets the MFMA read them there; its products run at
// ~12 cycles per instruction against the 8.3 of the plain rate test (mfma4x4_stream.hip, operands in architectural registers).
// Rate of 64 instructions per iteration, 4 accumulator chains, one wave per SIMD, with the B operand taken
//   V: from architectural registers,   A: straight from AGPRs,   R: from AGPRs through v_accvgpr_read a block ahead.
//   hipcc -O3 --offload-arch=gfx950 -Wno-unused-value -o mfma4x4_agpr tools/ubench/mfma4x4_agpr.hip && ./mfma4x4_agpr
// Measured (MI355X, round 5): V 8.31, A 8.72, R 16.06 cycles per instruction -- the AGPR-resident operands are NOT what costs
// k_solve_bcast its products' 12 cycles per instruction (a v_accvgpr_read in front of the MFMA, on the other hand, doubles them).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int J> __device__ __forceinline__ f32x4 mf(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 4, J, 0); }

template <int MODE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) k(float* out, const float* in, int iters, unsigned long long* stamp) {
    const int lane = threadIdx.x & 63;
    float RA[64];
#pragma unroll
    for (int r = 0; r < 64; ++r) {
        const float v = in[r * 64 + lane];
        if (MODE == 0) RA[r] = v;
        else asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(RA[r]) : "v"(v));
    }
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    float a0 = 0.001f * lane, a1 = 0.002f * lane;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int blk = 0; blk < 4; ++blk) {
            float bv[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                if (MODE == 2) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(bv[j]) : "a"(RA[16 * blk + j]));
                else bv[j] = RA[16 * blk + j];
            }
#define STEP(J) acc[(J) & 1] = mf<J>(a0, bv[J], acc[(J) & 1]); acc[2 + ((J) & 1)] = mf<J>(a1, bv[J], acc[2 + ((J) & 1)]);
            STEP(0) STEP(1) STEP(2) STEP(3) STEP(4) STEP(5) STEP(6) STEP(7) STEP(8) STEP(9) STEP(10) STEP(11) STEP(12) STEP(13) STEP(14) STEP(15)
        }
        a0 += 1e-6f; a1 -= 1e-6f;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + threadIdx.x] = acc[0].x + acc[1].y + acc[2].z + acc[3].w;
    if (threadIdx.x == 0) stamp[blockIdx.x] = t1 - t0;
}

template <int MODE> static void run(const char* what, float* dout, const float* din, unsigned long long* dst, int ncu) {
    const int iters = 2000;
    hipLaunchKernelGGL((k<MODE>), dim3(ncu), dim3(256), 0, 0, dout, din, 10, dst);
    hipLaunchKernelGGL((k<MODE>), dim3(ncu), dim3(256), 0, 0, dout, din, iters, dst);
    hipDeviceSynchronize();
    std::vector<unsigned long long> st(ncu);
    hipMemcpy(st.data(), dst, ncu * 8, hipMemcpyDeviceToHost);
    double c = 0; for (int i = 0; i < ncu; ++i) c += st[i];
    printf("B operand %s: %.2f cycles per v_mfma_f32_4x4x1_16B_f32 (4 chains, one wave per SIMD, all CUs busy)\n", what, c / ncu / (iters * 128.0));
}
int main() {
    int ncu = 0; hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
    float *dout, *din; unsigned long long* dst;
    hipMalloc(&dout, (size_t)ncu * 256 * 4); hipMalloc(&din, 64 * 64 * 4); hipMalloc(&dst, ncu * 8);
    std::vector<float> h(64 * 64); for (auto& v : h) v = (float)rand() / RAND_MAX - 0.5f;
    hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    run<0>("from architectural VGPRs", dout, din, dst, ncu);
    run<1>("straight from AGPRs", dout, din, dst, ncu);
    run<2>("from AGPRs through v_accvgpr_read, a block of 16 ahead", dout, din, dst, ncu);
    return 0;
}
