// Microbenchmark behind k_solve_bcast (cnf_bcast.hip): is an 8-samples-per-CU formulation of BASELINE config 5 (128-384-128,
// B = 2048 = 8 samples per CU) feasible at the fp32 MFMA rate?
//
// v_mfma_f32_4x4x1_16B_f32 multiplies 16 independent 4x4x1 blocks per instruction (8 cycles, 64 flop/clk/SIMD: the rate of
// every fp32 MFMA).  With CBSZ = 4 the A operand of block ABID is BROADCAST to all 16 blocks, so with A = activations
// (4 samples x 1 feature k) and B = weights (lane = output feature, 64 per instruction) one instruction forms
// D[sample 0..3 in the 4 registers][feature = lane] += x[k][sample] * W[feature][k]: a 64-feature x 4-sample x K = 1 product
// with NO lane wasted on samples that do not exist -- the 16x16x4 form needs 16 samples per wave.  One A register carries 16
// k's (lane 4j + s = x[k0 + j][s]; ABID = j picks one).  The price: every weight serves only 4 (x sample groups) samples per
// fetch, so the weights stream from L2 at the CU's full 64 B/clk while the matrix pipe runs.
//
// Part 1 checks the lane maps with exact integer data (A: lane 4b + i = row i of block b; B: lane 4b + j = column j of block
// b; D: register i, lane 4b + j; CBSZ / ABID as above).  Part 2 times the streaming loop: per "evaluation" every CU reads
// 786 KB of fp32 weights (four 49152-float images, fragment-ordered) with 4 waves, each 16-byte load feeding 8 MFMAs (4 k's x
// 2 sample groups): 6144 MFMAs per CU = 12.3 k cycles at the pipe's rate, 12.3 k cycles of L2 stream at 64 B/clk.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/mfma4x4_stream.hip -o /tmp/m4 && /tmp/m4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ABID is an instruction field: a literal at every call site (the switch folds once its argument is a constant after unrolling)
__device__ __forceinline__ f32x4 mfma_bc(float a, float b, f32x4 c, int abid) {
    switch (abid & 15) {
#define MB_CASE(J) case J: return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 4, J, 0);
        MB_CASE(0) MB_CASE(1) MB_CASE(2) MB_CASE(3) MB_CASE(4) MB_CASE(5) MB_CASE(6) MB_CASE(7)
        MB_CASE(8) MB_CASE(9) MB_CASE(10) MB_CASE(11) MB_CASE(12) MB_CASE(13) MB_CASE(14) default: return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 4, 15, 0);
#undef MB_CASE
    }
}

__global__ void k_layout(const float* A, const float* B, float* D, int abid_mode) {
    const int lane = threadIdx.x;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (abid_mode == 0) acc = __builtin_amdgcn_mfma_f32_4x4x1f32(A[lane], B[lane], acc, 0, 0, 0);
    else if (abid_mode == 1) acc = __builtin_amdgcn_mfma_f32_4x4x1f32(A[lane], B[lane], acc, 4, 0, 0);
    else if (abid_mode == 2) acc = __builtin_amdgcn_mfma_f32_4x4x1f32(A[lane], B[lane], acc, 4, 5, 0);
    else acc = __builtin_amdgcn_mfma_f32_4x4x1f32(A[lane], B[lane], acc, 4, 15, 0);
    for (int r = 0; r < 4; ++r) D[r * 64 + lane] = acc[r];
}

template <int NSG, int PF>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
k_stream(const float* __restrict__ img, float* out, int evals, unsigned long long* stamp) {
    __shared__ float xa[NSG][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < NSG * 64; i += 256) (&xa[0][0])[i] = 0.001f * (i % 17);
    __syncthreads();
    constexpr int NLOAD = 192;                             // 16-byte loads per lane per wave and evaluation: 196608 floats / 4 waves / 64 lanes / 4
    const f32x4* src = reinterpret_cast<const f32x4*>(img) + (size_t)wave * NLOAD * 64 + lane;
    f32x4 acc[NSG];
    for (int s = 0; s < NSG; ++s) acc[s] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int e = 0; e < evals; ++e) {
        f32x4 buf[PF];
#pragma unroll
        for (int p = 0; p < PF; ++p) buf[p] = src[(size_t)p * 64];
        for (int i0 = 0; i0 < NLOAD; i0 += PF) {
            float a[NSG];
#pragma unroll
            for (int s = 0; s < NSG; ++s) a[s] = xa[s][lane] + (float)(i0 & 3);
#pragma unroll
            for (int p = 0; p < PF; ++p) {
                const f32x4 b = buf[p];
                const int nx = i0 + p + PF;
                buf[p] = src[(size_t)(nx < NLOAD ? nx : NLOAD - 1) * 64];
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int s = 0; s < NSG; ++s)
                        acc[s] = mfma_bc(a[s], b[c], acc[s], 4 * (p & 3) + c);
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float r = 0.f;
    for (int s = 0; s < NSG; ++s) r += acc[s].x + acc[s].y + acc[s].z + acc[s].w;
    out[blockIdx.x * 256 + threadIdx.x] = r;
    if (threadIdx.x == 0) { stamp[2 * blockIdx.x] = t1 - t0; stamp[2 * blockIdx.x + 1] = r1 - r0; }
}

// the matrix pipe alone: NACC independent accumulators, operands in registers, no memory traffic
template <int NACC, bool BCAST>
__global__ void __launch_bounds__(256) k_rate(float* out, int iters, unsigned long long* stamp) {
    const int lane = threadIdx.x & 63;
    f32x4 acc[NACC];
    for (int s = 0; s < NACC; ++s) acc[s] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = 0.001f * lane, b[4] = {1.f + lane, 2.f, 0.5f, 0.25f};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int s = 0; s < NACC; ++s)
                acc[s] = BCAST ? mfma_bc(a, b[u & 3], acc[s], u) : __builtin_amdgcn_mfma_f32_4x4x1f32(a, b[u & 3], acc[s], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0.f;
    for (int s = 0; s < NACC; ++s) r += acc[s].x + acc[s].w;
    out[blockIdx.x * 256 + threadIdx.x] = r;
    if (threadIdx.x == 0) stamp[blockIdx.x] = t1 - t0;
}
template <int NACC, bool BCAST>
static void rate(float* dout, unsigned long long* dst, int ncu) {
    const int iters = 2000;
    hipLaunchKernelGGL((k_rate<NACC, BCAST>), dim3(ncu), dim3(256), 0, 0, dout, iters, dst);
    hipDeviceSynchronize();
    std::vector<unsigned long long> st(ncu);
    hipMemcpy(st.data(), dst, st.size() * 8, hipMemcpyDeviceToHost);
    double cyc = 0;
    for (int i = 0; i < ncu; ++i) cyc += st[i];
    printf("v_mfma_f32_4x4x1_16B_f32%s, %d accumulators, one wave per SIMD: %.2f cycles per instruction\n", BCAST ? " (CBSZ=4)" : "", NACC,
           cyc / ncu / ((double)iters * 16 * NACC));
}

template <int NSG, int PF>
static void run(const float* dimg, float* dout, unsigned long long* dst, int ncu) {
    const int evals = 200;
    hipLaunchKernelGGL((k_stream<NSG, PF>), dim3(ncu), dim3(256), 0, 0, dimg, dout, 3, dst);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_stream<NSG, PF>), dim3(ncu), dim3(256), 0, 0, dimg, dout, evals, dst);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> st(2 * ncu);
    hipMemcpy(st.data(), dst, st.size() * 8, hipMemcpyDeviceToHost);
    double cyc = 0, rt = 0;
    for (int i = 0; i < ncu; ++i) { cyc += st[2 * i]; rt += st[2 * i + 1]; }
    cyc /= ncu; rt /= ncu;
    const double us_eval = ms * 1e3 / evals;
    const double flops = (double)ncu * 4 * 192 * 4 * NSG * 512.0;        // per evaluation
    printf("NSG=%d (samples per CU %2d) prefetch %2d loads: %.2f us per evaluation-equivalent (%.1f k cycles at %.2f GHz), %.1f TFLOP/s fp32 "
           "= %.2f of 157.3, weight stream %.0f GB/s per CU (%.1f TB/s chip)\n",
           NSG, 4 * NSG, PF, us_eval, cyc / evals / 1e3, cyc / rt / 10.0, flops / us_eval / 1e6, flops / us_eval / 1e6 / 157.3,
           786432.0 / us_eval / 1e3, 786432.0 * ncu / us_eval / 1e6);
}

int main() {
    // ---- part 1: lane maps ----
    float hA[64], hB[64], hD[256], *dA, *dB, *dD;
    for (int l = 0; l < 64; ++l) { hA[l] = (float)(1 + l); hB[l] = (float)(100 + 3 * l); }
    hipMalloc(&dA, 256); hipMalloc(&dB, 256); hipMalloc(&dD, 1024);
    hipMemcpy(dA, hA, 256, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 256, hipMemcpyHostToDevice);
    int bad = 0;
    for (int mode = 0; mode < 4; ++mode) {
        hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, dA, dB, dD, mode);
        hipMemcpy(hD, dD, 1024, hipMemcpyDeviceToHost);
        const int ab = mode == 2 ? 5 : (mode == 3 ? 15 : 0);
        for (int r = 0; r < 4; ++r)
            for (int l = 0; l < 64; ++l) {
                const int b = l / 4;
                const int ablk = mode == 0 ? b : ab;                      // whose A this block sees
                const float want = hA[4 * ablk + r] * hB[l];              // D[i = r][j = l % 4] of block b = A[i] * B[j]
                if (hD[r * 64 + l] != want) { if (bad < 5) printf("mode %d reg %d lane %d: got %g want %g\n", mode, r, l, hD[r * 64 + l], want); ++bad; }
            }
    }
    printf("lane maps (A: lane 4b+i, B: lane 4b+j, D: reg i lane 4b+j; CBSZ=4 broadcasts block ABID's A): %s\n", bad ? "MISMATCH" : "ok");
    // ---- part 2: the streaming loop ----
    int ncu = 0;
    hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
    const size_t nimg = 196608;
    std::vector<float> h(nimg);
    for (size_t i = 0; i < nimg; ++i) h[i] = (float)((double)rand() / RAND_MAX - 0.5);
    float *dimg, *dout; unsigned long long* dst;
    hipMalloc(&dimg, nimg * 4); hipMalloc(&dout, (size_t)ncu * 256 * 4); hipMalloc(&dst, (size_t)ncu * 16);
    hipMemcpy(dimg, h.data(), nimg * 4, hipMemcpyHostToDevice);
    rate<1, true>(dout, dst, ncu); rate<2, true>(dout, dst, ncu); rate<4, true>(dout, dst, ncu); rate<8, true>(dout, dst, ncu);
    rate<2, false>(dout, dst, ncu); rate<8, false>(dout, dst, ncu);
    run<2, 8>(dimg, dout, dst, ncu);
    run<2, 12>(dimg, dout, dst, ncu);
    run<2, 16>(dimg, dout, dst, ncu);
    run<2, 24>(dimg, dout, dst, ncu);
    run<1, 16>(dimg, dout, dst, ncu);
    run<4, 16>(dimg, dout, dst, ncu);
    run<2, 16>(dimg, dout, dst, ncu / 2);
    return bad != 0;
}
