// Repro for the exit-time SIGSEGV seen twice under `rocprofv3 -- python3 bench.py` when the one-launch solve went through
// hipLaunchCooperativeKernel (gpurun_out/tl2.log of round 2, prof_pers.log of round 1): a process whose only GPU work is
// one trivial kernel, launched cooperatively (argv[1] = "coop") or ordinarily ("plain"), then a normal exit().  An atexit
// hook registered FIRST (so it runs last) writes /proc/self/maps next to the profile, so that the frames of a crash inside
// exit() can be attributed to libraries.  Build: hipcc --offload-arch=gfx950 -o coop_exit tools/ubench/coop_exit.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>

static const char* g_maps_out = nullptr;
static void dump_maps() {
    if (!g_maps_out) return;
    FILE* in = fopen("/proc/self/maps", "r");
    FILE* out = fopen(g_maps_out, "w");
    if (in && out) {
        char buf[4096];
        size_t n;
        while ((n = fread(buf, 1, sizeof buf, in)) > 0) fwrite(buf, 1, n, out);
    }
    if (in) fclose(in);
    if (out) fclose(out);
}

__global__ void k_touch(float* p) { p[threadIdx.x + blockIdx.x * blockDim.x] += 1.f; }

int main(int argc, char** argv) {
    const bool coop = argc > 1 && !strcmp(argv[1], "coop");
    g_maps_out = argc > 2 ? argv[2] : nullptr;
    atexit(dump_maps);                      // registered before the HIP runtime registers anything: runs after all of it
    float* d = nullptr;
    if (hipMalloc(&d, 256 * 512 * sizeof(float)) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); return 2; }
    (void)hipMemset(d, 0, 256 * 512 * sizeof(float));
    void* args[] = {&d};
    hipError_t e = coop ? hipLaunchCooperativeKernel((const void*)k_touch, dim3(256), dim3(512), args, 0, nullptr)
                        : hipLaunchKernel((const void*)k_touch, dim3(256), dim3(512), args, 0, nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    float v = 0.f;
    if (e == hipSuccess) e = hipMemcpy(&v, d, sizeof v, hipMemcpyDeviceToHost);
    printf("%s launch: %s, p[0] = %g\n", coop ? "cooperative" : "ordinary", hipGetErrorString(e), v);
    (void)hipFree(d);
    return e == hipSuccess ? 0 : 1;
}
