// Which lanes of a wave share a pass of ds_write_b64 / ds_read_b128 on gfx950?  Every bank pair (8 bytes of a 256-byte LDS
// row) is given to exactly TWO lanes, L and L ^ mask, in different rows: a 64-lane b64 store touches 512 bytes, so two
// passes are the minimum; if the hardware puts L and L ^ mask into the SAME pass they conflict and the store takes longer.
// Prints cycles per instruction (s_memtime, one wave, 4096 stores) for every mask, and the same for the epilogue-store
// pattern of k_step3b (mask 33).  Build: hipcc --offload-arch=gfx950 -O3 -o lds_b64 tools/ubench/lds_b64.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ void k_b64(const int* __restrict__ addr, unsigned long long* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int a = addr[threadIdx.x];
    unsigned long long t0 = 0, t1 = 0;
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    u32x2 v = {threadIdx.x, 7u};
    __syncthreads();
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            *(u32x2*)(lds + a + 512 * 0) = v;      // (same addresses every time: only the lane -> bank map matters)
            asm volatile("" ::: "memory");
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = *(unsigned*)(lds + a); }
}

int main() {
    int* d_addr; unsigned long long* d_out;
    hipMalloc(&d_addr, 64 * sizeof(int)); hipMalloc(&d_out, 16);
    const int iters = 512;
    auto run = [&](const char* name, const int* h) {
        hipMemcpy(d_addr, h, 64 * sizeof(int), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_b64, dim3(1), dim3(64), 8192, 0, d_addr, d_out, iters);
        hipLaunchKernelGGL(k_b64, dim3(1), dim3(64), 8192, 0, d_addr, d_out, iters);
        unsigned long long r[2];
        hipMemcpy(r, d_out, 16, hipMemcpyDeviceToHost);
        printf("%-28s %6.2f cycles per ds_write_b64\n", name, (double)r[0] / (iters * 8.0));
    };
    int h[64];
    for (int L = 0; L < 64; ++L) h[L] = 8 * L;
    run("linear (8 L)", h);
    const int masks[] = {1, 2, 4, 8, 16, 32, 33, 3, 17, 48, 34, 36, 40, 63};
    for (int mask : masks) {
        // pair index: rank of min(L, L ^ mask) among the pair leaders
        int leader_rank[64], rk = 0;
        for (int L = 0; L < 64; ++L) if (L < (L ^ mask)) leader_rank[L] = rk++;
        for (int L = 0; L < 64; ++L) {
            const int P = L ^ mask, lead = L < P ? L : P;
            h[L] = 8 * leader_rank[lead] + 256 * (L < P ? 0 : 1);
        }
        char name[64]; snprintf(name, sizeof name, "partner = L ^ %d", mask);
        run(name, h);
    }
    // k_step3b's epilogue store: lane (q, s) of wave w -> row s, chunk (2w + (q >> 1)) ^ s, half q & 1
    for (int w = 0; w < 2; ++w) {
        for (int L = 0; L < 64; ++L) { const int s = L & 15, q = L >> 4; h[L] = s * 256 + 16 * ((2 * w + (q >> 1)) ^ s) + 8 * (q & 1); }
        char name[64]; snprintf(name, sizeof name, "k_step3b store, wave %d", w);
        run(name, h);
    }
    return 0;
}
