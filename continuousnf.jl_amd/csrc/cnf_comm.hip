// RCCL side of the C ABI (SURVEY.md section 8 b'/e): the one collective of the path, the all-reduce of the five
// loss sums behind `mean` in loss (src/icnf.jl:489, src/base_icnf.jl:496), callable without torch -- a Julia
// caller creates the communicator and reduces through these entry points.
//
// librccl is resolved at first use with dlopen/dlsym, not linked: a process that already holds an RCCL (PyTorch
// ships its own copy next to its HIP runtime) keeps using that one, a Julia process gets the system library,
// and libcnfhip.so still loads on a machine without RCCL (single-GPU use).
#include "../../include/cnfhip.h"
#include <hip/hip_runtime.h>

#include <dlfcn.h>
#include <unistd.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>

namespace {

struct NcclId { char internal[CNF_COMM_ID_BYTES]; };       // ncclUniqueId (rccl.h: 128 opaque bytes), passed by value
typedef int (*fn_get_unique_id)(NcclId*);
typedef int (*fn_comm_init_rank)(void**, int, NcclId, int);
typedef int (*fn_all_reduce)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*fn_comm_destroy)(void*);
typedef int (*fn_comm_count)(void*, int*);
typedef const char* (*fn_error_string)(int);
const int kNcclFloat32 = 7, kNcclSum = 0;                  // rccl.h: ncclFloat32 = 7, ncclSum = 0

struct Rccl {
    void* lib = nullptr;
    fn_get_unique_id get_unique_id = nullptr;
    fn_comm_init_rank comm_init_rank = nullptr;
    fn_all_reduce all_reduce = nullptr;
    fn_comm_destroy comm_destroy = nullptr;
    fn_comm_count comm_count = nullptr;
    fn_error_string error_string = nullptr;
    std::string path;
};
Rccl g_rccl;
std::once_flag g_once;
thread_local std::string t_err;

void load_rccl() {
    const char* env = getenv("CNFHIP_RCCL_LIB");
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    void* lib = nullptr;
    std::string used;
    if (env && env[0]) { lib = dlopen(env, RTLD_NOW | RTLD_GLOBAL); used = env; }
    for (int pass = 0; pass < 2 && !lib; ++pass)           // pass 0: whatever the process already holds
        for (const char* n : names) {
            lib = dlopen(n, pass == 0 ? (RTLD_NOW | RTLD_NOLOAD) : (RTLD_NOW | RTLD_GLOBAL));
            if (lib) { used = n; break; }
        }
    if (!lib) {
        // a copy loaded under another name (torch/lib/librccl.so): its symbols are visible process-wide
        if (dlsym(RTLD_DEFAULT, "ncclAllReduce")) { lib = RTLD_DEFAULT; used = "(process image)"; }
        else return;
    }
    g_rccl.get_unique_id = (fn_get_unique_id)dlsym(lib, "ncclGetUniqueId");
    g_rccl.comm_init_rank = (fn_comm_init_rank)dlsym(lib, "ncclCommInitRank");
    g_rccl.all_reduce = (fn_all_reduce)dlsym(lib, "ncclAllReduce");
    g_rccl.comm_destroy = (fn_comm_destroy)dlsym(lib, "ncclCommDestroy");
    g_rccl.comm_count = (fn_comm_count)dlsym(lib, "ncclCommCount");
    g_rccl.error_string = (fn_error_string)dlsym(lib, "ncclGetErrorString");
    if (g_rccl.get_unique_id && g_rccl.comm_init_rank && g_rccl.all_reduce && g_rccl.comm_destroy) {
        g_rccl.lib = lib;
        g_rccl.path = used;
    }
}

const Rccl* rccl() {
    std::call_once(g_once, load_rccl);
    if (!g_rccl.lib) { t_err = "librccl could not be loaded (set CNFHIP_RCCL_LIB to its path)"; return nullptr; }
    return &g_rccl;
}

cnf_status rccl_fail(const Rccl* r, const char* what, int code) {
    char buf[256];
    snprintf(buf, sizeof buf, "%s failed: %s (ncclResult %d)", what,
             r->error_string ? r->error_string(code) : "?", code);
    t_err = buf;
    return CNF_ERR_RCCL;
}

}  // namespace

extern "C" const char* cnf_comm_last_error(void) { return t_err.c_str(); }

extern "C" const char* cnf_comm_library(void) {
    const Rccl* r = rccl();
    return r ? r->path.c_str() : "";
}

// "<hostname>/<pci bus id>" of a device: what the ranks exchange BEFORE cnf_comm_init to make sure that no two of them sit
// on one GPU.  RCCL 2.26 answers that with ncclInvalidUsage ("Duplicate GPU detected", gpurun_out/rccl2.log) from inside
// ncclCommInitRank -- after its own bootstrap, on every rank, with the failed communicator's resources to clean up.  Settling it
// in the caller's bootstrap, where the ranks can still talk, gives every rank the same plain answer before RCCL is entered.
extern "C" cnf_status cnf_comm_device_key(int device, char* out, size_t cap) {
    if (!out || cap < 2) return CNF_ERR_BAD_ARG;
    out[0] = 0;
    int dev = device;
    if (dev < 0 && hipGetDevice(&dev) != hipSuccess) { t_err = "no current device"; return CNF_ERR_NO_DEVICE; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { t_err = "no gfx950 device"; return CNF_ERR_NO_DEVICE; }
    if (dev >= ndev) { t_err = "device ordinal out of range"; return CNF_ERR_BAD_ARG; }
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, dev) != hipSuccess) { t_err = "hipDeviceGetPCIBusId failed"; return CNF_ERR_HIP; }
    char host[128] = {0};
    if (gethostname(host, sizeof host - 1) != 0) snprintf(host, sizeof host, "?");
    // (two containers of one machine may share a hostname or not; the boot id tells machines apart where it is readable)
    char boot[64] = {0};
    if (FILE* f = fopen("/proc/sys/kernel/random/boot_id", "r")) {
        if (!fgets(boot, sizeof boot, f)) boot[0] = 0;
        fclose(f);
        for (char* c = boot; *c; ++c) if (*c == '\n') *c = 0;
    }
    const int n = snprintf(out, cap, "%s|%s/%s", host, boot, bus);
    if (n < 0 || (size_t)n >= cap) { t_err = "key buffer too small"; return CNF_ERR_BAD_ARG; }
    return CNF_OK;
}

extern "C" cnf_status cnf_comm_unique_id(char* id) {
    if (!id) return CNF_ERR_BAD_ARG;
    const Rccl* r = rccl();
    if (!r) return CNF_ERR_RCCL;
    NcclId nid;
    memset(&nid, 0, sizeof nid);
    const int rc = r->get_unique_id(&nid);
    if (rc != 0) return rccl_fail(r, "ncclGetUniqueId", rc);
    memcpy(id, nid.internal, CNF_COMM_ID_BYTES);
    return CNF_OK;
}

extern "C" cnf_status cnf_comm_init(cnf_comm* out, int world_size, int rank, const char* id, int device) {
    if (!out || !id) return CNF_ERR_BAD_ARG;
    *out = nullptr;
    if (world_size < 1 || rank < 0 || rank >= world_size) { t_err = "bad rank / world size"; return CNF_ERR_BAD_ARG; }
    const Rccl* r = rccl();
    if (!r) return CNF_ERR_RCCL;
    if (device >= 0) {                   // device < 0: the caller has already selected the device of this rank
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { t_err = "no gfx950 device"; return CNF_ERR_NO_DEVICE; }
        if (device >= ndev) { t_err = "device ordinal out of range"; return CNF_ERR_BAD_ARG; }
        if (hipSetDevice(device) != hipSuccess) { t_err = "hipSetDevice failed"; return CNF_ERR_HIP; }
    }
    NcclId nid;
    memcpy(nid.internal, id, CNF_COMM_ID_BYTES);
    void* comm = nullptr;
    const int rc = r->comm_init_rank(&comm, world_size, nid, rank);
    if (rc != 0) return rccl_fail(r, "ncclCommInitRank", rc);
    *out = comm;
    return CNF_OK;
}

extern "C" cnf_status cnf_comm_destroy(cnf_comm comm) {
    if (!comm) return CNF_ERR_BAD_ARG;
    const Rccl* r = rccl();
    if (!r) return CNF_ERR_RCCL;
    const int rc = r->comm_destroy(comm);
    return rc == 0 ? CNF_OK : rccl_fail(r, "ncclCommDestroy", rc);
}

extern "C" cnf_status cnf_comm_size(cnf_comm comm, int* world_size) {
    if (!comm || !world_size) return CNF_ERR_BAD_ARG;
    const Rccl* r = rccl();
    if (!r) return CNF_ERR_RCCL;
    if (!r->comm_count) { t_err = "ncclCommCount not exported"; return CNF_ERR_RCCL; }
    const int rc = r->comm_count(comm, world_size);
    return rc == 0 ? CNF_OK : rccl_fail(r, "ncclCommCount", rc);
}

// In-place ncclAllReduce(ncclSum, ncclFloat32) of `n` device floats on `stream`: the 5 loss sums, the 3 floats
// of the lock-step controller, or n_params + 2 floats of a data-parallel gradient.
extern "C" cnf_status cnf_comm_allreduce(cnf_comm comm, float* buf_dev, size_t n, void* stream) {
    if (!comm || !buf_dev) return CNF_ERR_BAD_ARG;
    if (n == 0) return CNF_OK;
    const Rccl* r = rccl();
    if (!r) return CNF_ERR_RCCL;
    const int rc = r->all_reduce(buf_dev, buf_dev, n, kNcclFloat32, kNcclSum, comm, (hipStream_t)stream);
    return rc == 0 ? CNF_OK : rccl_fail(r, "ncclAllReduce", rc);
}
