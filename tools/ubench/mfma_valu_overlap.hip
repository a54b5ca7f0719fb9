// Do a wave's own vector instructions hide under its matrix instructions better with ONE wave per SIMD than the work of two
// SIMD partners hides under each other?  Per iteration and SIMD the same work in every variant: 48 six-term-style bf16
// MFMAs (v_mfma_f32_16x16x32_bf16, four independent accumulator chains) and the epilogue of 16 accumulator elements per
// lane (tanh by exp2 / rcp, sigma', the three-piece bf16 split: ~14 vector instructions each).
//   W = 4 waves (one per SIMD, each does all of it)  |  W = 8 waves (two per SIMD, each does half)
//   SEQ: products, then the epilogue of THEIR results        INT: products of iteration i beside the epilogue of i - 1
// Prints cycles per iteration (s_memtime of wave 0).  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
typedef float f32x2_ __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned epi2(float v0, float v1, float& keep) {
    // tanh, sigma', split of two elements -> one packed word of each piece folded together (keeps everything alive)
    const float t0 = __builtin_amdgcn_exp2f(v0 * 2.885390f), t1 = __builtin_amdgcn_exp2f(v1 * 2.885390f);
    const float h0 = fmaf(-2.f, __builtin_amdgcn_rcpf(t0 + 1.f), 1.f), h1 = fmaf(-2.f, __builtin_amdgcn_rcpf(t1 + 1.f), 1.f);
    keep += fmaf(-h0, h0, 1.f) + fmaf(-h1, h1, 1.f);
    const unsigned hh = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_{h0, h1}, bf16x2_));
    const float r0 = h0 - __uint_as_float(hh << 16), r1 = h1 - __uint_as_float(hh & 0xffff0000u);
    const unsigned mm = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_{r0, r1}, bf16x2_));
    const float s0 = r0 - __uint_as_float(mm << 16), s1 = r1 - __uint_as_float(mm & 0xffff0000u);
    const unsigned ll = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_{s0, s1}, bf16x2_));
    return hh ^ mm ^ ll;
}

template <int NM, bool INT, int WHAT = 3>      // NM = MFMAs per wave and iteration (4 chains); epilogue of NM / 3 elements per lane; WHAT: 1 products only, 2 epilogue only
__global__ void k(unsigned long long* out, int iters, float seed) {
    const int lane = threadIdx.x & 63;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed * (lane + i)); b[i] = (__bf16)(seed * (lane - i)); }
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}}, prev[4] = {{.1f, .2f, .3f, .4f}, {.1f, .2f, .3f, .4f}, {.1f, .2f, .3f, .4f}, {.1f, .2f, .3f, .4f}};
    float keep = 0.f; unsigned xw = 0;
    constexpr int PER = NM / 4;                       // MFMAs per chain
    constexpr int NE = NM / 3;                        // elements per lane in the epilogue (16 for NM = 48)
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (INT) {
            // products of this iteration with the epilogue of the previous one's results spread between them
#pragma unroll
            for (int j = 0; j < PER; ++j) {
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[c], 0, 0, 0);
                if (j < NE / 2) { const int e = 2 * j; xw ^= epi2(prev[(e >> 2) & 3][e & 3], prev[((e + 1) >> 2) & 3][(e + 1) & 3], keep); }
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) { prev[c] = acc[c] * 1e-3f; acc[c] = f32x4{0, 0, 0, 0}; }
        } else {
            if (WHAT & 1) {
#pragma unroll
            for (int j = 0; j < PER; ++j)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[c], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (WHAT & 2) {
#pragma unroll
            for (int e = 0; e < NE; e += 2) xw ^= epi2(acc[(e >> 2) & 3][e & 3] * 1e-3f + prev[0][e & 3] + (float)it * 1e-9f, acc[((e + 1) >> 2) & 3][(e + 1) & 3] * 1e-3f + (float)it * 1e-9f, keep);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (WHAT == 3) {
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] = f32x4{0, 0, 0, 0};
            } else if (WHAT == 1) {
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] *= 1e-3f;            // (keeps the chains bounded and alive)
            }
        }
    }
    keep += acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = (unsigned long long)(keep + (float)xw + prev[0][0]); }
}

int main() {
    unsigned long long* d; hipMalloc(&d, 64);
    const int iters = 2000;
    auto report = [&](const char* name) {
        unsigned long long r[2]; hipMemcpy(r, d, 16, hipMemcpyDeviceToHost);
        printf("%-46s %8.1f cycles per iteration (per SIMD: 48 MFMAs = 768 pipe cycles + 16 elements x 2 waves-worth of epilogue)\n", name, (double)r[0] / iters);
    };
    hipLaunchKernelGGL((k<48, false, 1>), dim3(1), dim3(256), 0, 0, d, iters, 0.01f); report("4 waves, products only");
    hipLaunchKernelGGL((k<48, false, 2>), dim3(1), dim3(256), 0, 0, d, iters, 0.01f); report("4 waves, epilogue only");
    hipLaunchKernelGGL((k<24, false, 1>), dim3(1), dim3(512), 0, 0, d, iters, 0.01f); report("8 waves, products only");
    hipLaunchKernelGGL((k<24, false, 2>), dim3(1), dim3(512), 0, 0, d, iters, 0.01f); report("8 waves, epilogue only");
    hipLaunchKernelGGL((k<48, false>), dim3(1), dim3(256), 0, 0, d, iters, 0.01f); report("4 waves, products then epilogue");
    hipLaunchKernelGGL((k<48, true>), dim3(1), dim3(256), 0, 0, d, iters, 0.01f); report("4 waves, epilogue(i-1) between products(i)");
    hipLaunchKernelGGL((k<24, false>), dim3(1), dim3(512), 0, 0, d, iters, 0.01f); report("8 waves, products then epilogue");
    hipLaunchKernelGGL((k<24, true>), dim3(1), dim3(512), 0, 0, d, iters, 0.01f); report("8 waves, epilogue(i-1) between products(i)");
    return 0;
}
