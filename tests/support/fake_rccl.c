/* Test double for librccl (tests only; never shipped, never used on a GPU box).
 *
 * libcnfhip.so resolves RCCL with dlopen (csrc/cnf_comm.hip); with CNFHIP_RCCL_LIB pointing here the
 * world-size-2 CPU test drives the real cnf_comm_* entry points end to end -- symbol lookup, the
 * by-value 128-byte ncclUniqueId, the ncclFloat32 / ncclSum enum values, argument order -- without a
 * GPU.  The "collective" is a sum over the ranks through a shared file in /dev/shm; buffers are host
 * memory here.  Signatures follow /opt/rocm/include/rccl/rccl.h. */
#define _GNU_SOURCE
#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>

#define ID_BYTES 128
#define MAX_RANKS 8
#define MAX_COUNT 65536
typedef struct { char internal[ID_BYTES]; } ncclUniqueId;
typedef struct {
    volatile uint32_t arrive;      /* monotonic arrival counter */
    volatile uint32_t leave;
    float slot[MAX_RANKS][MAX_COUNT];
} shm_t;
typedef struct { shm_t* shm; int nranks, rank; uint32_t gen; char name[ID_BYTES]; } comm_t;

int ncclGetUniqueId(ncclUniqueId* id) {
    memset(id, 0, sizeof *id);
    struct timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);
    snprintf(id->internal, ID_BYTES, "/fakerccl_%d_%ld", (int)getpid(), (long)ts.tv_nsec);
    return 0;
}

int ncclCommInitRank(void** out, int nranks, ncclUniqueId id, int rank) {
    if (nranks < 1 || nranks > MAX_RANKS || rank < 0 || rank >= nranks) return 4;   /* ncclInvalidArgument */
    if (strncmp(id.internal, "/fakerccl_", 10) != 0) return 4;                      /* id did not arrive intact */
    int fd = shm_open(id.internal, O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, sizeof(shm_t)) != 0) return 2;                       /* ncclSystemError */
    shm_t* p = (shm_t*)mmap(NULL, sizeof(shm_t), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return 2;
    comm_t* c = (comm_t*)calloc(1, sizeof *c);
    c->shm = p; c->nranks = nranks; c->rank = rank; c->gen = 0;
    memcpy(c->name, id.internal, ID_BYTES);
    *out = c;
    return 0;
}

static int wait_for(volatile uint32_t* w, uint32_t target) {
    for (long spins = 0; (int32_t)(__atomic_load_n(w, __ATOMIC_ACQUIRE) - target) < 0; ++spins) {
        if (spins > 2000000000L) return 1;
        if ((spins & 1023) == 1023) usleep(50);
    }
    return 0;
}

int ncclAllReduce(const void* send, void* recv, size_t count, int dtype, int op, void* comm, void* stream) {
    (void)stream;
    comm_t* c = (comm_t*)comm;
    if (!c || !send || !recv) return 4;
    if (dtype != 7 || op != 0) return 4;            /* ncclFloat32, ncclSum only */
    if (count > MAX_COUNT) return 4;
    memcpy((void*)c->shm->slot[c->rank], send, count * sizeof(float));
    c->gen += 1;
    __atomic_fetch_add(&c->shm->arrive, 1u, __ATOMIC_ACQ_REL);
    if (wait_for(&c->shm->arrive, c->gen * (uint32_t)c->nranks)) return 3;
    float* out = (float*)recv;
    for (size_t i = 0; i < count; ++i) {
        float s = 0.f;
        for (int r = 0; r < c->nranks; ++r) s += c->shm->slot[r][i];       /* rank order: same sum everywhere */
        out[i] = s;
    }
    __atomic_fetch_add(&c->shm->leave, 1u, __ATOMIC_ACQ_REL);
    if (wait_for(&c->shm->leave, c->gen * (uint32_t)c->nranks)) return 3;   /* slots free for the next call */
    return 0;
}

int ncclCommCount(void* comm, int* n) { *n = ((comm_t*)comm)->nranks; return 0; }

int ncclCommDestroy(void* comm) {
    comm_t* c = (comm_t*)comm;
    munmap(c->shm, sizeof(shm_t));
    if (c->rank == 0) shm_unlink(c->name);
    free(c);
    return 0;
}

const char* ncclGetErrorString(int r) {
    switch (r) { case 0: return "no error"; case 2: return "system error"; case 3: return "internal error";
                 case 4: return "invalid argument"; default: return "unknown"; }
}
