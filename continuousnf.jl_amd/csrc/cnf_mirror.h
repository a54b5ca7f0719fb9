// Host mirror of the integrator state for the streamed solve (cnf_abi.hip): block 0 of every step launch copies the
// state it has just computed into pinned, host-coherent memory and the host polls it instead of waiting on events.
//
// Protocol: the payload carries its own sequence number.  Every 32-bit word of the state travels as one naturally
// aligned 8-byte granule {tag = launch index, word}, written by ONE 8-byte store; the reader takes a snapshot only
// if all granules carry the same tag.  No ordering between the stores is assumed (none is guaranteed for relaxed
// system-scope stores over PCIe), the writer never waits for its stores, and a snapshot torn between two launches
// can not be accepted: some granule then carries the other tag.  (A one-word "state, then index" hand-off is not a
// seqlock: a reader descheduled between its two index reads could accept words of two launches.)
//
// Plain C++ (no HIP): tests/support/mirror_test.cpp runs the reader against a writer thread on the CPU.
#pragma once
#include <cstdint>
#include <cstring>

template <class State>
struct CnfMirrorT {
    static_assert(sizeof(State) % 4 == 0, "copied as 32-bit words");
    static constexpr int kWords = (int)(sizeof(State) / 4);
    uint64_t g[kWords];
};

// One pass over the granules.  Returns true and fills *out / *tag iff all granules carry the same tag.
template <class State>
static inline bool cnf_mirror_read(const volatile CnfMirrorT<State>* m, State* out, uint32_t* tag) {
    constexpr int N = CnfMirrorT<State>::kWords;
    uint32_t words[N];
    const uint64_t g0 = __atomic_load_n(&m->g[0], __ATOMIC_RELAXED);
    const uint32_t t = (uint32_t)(g0 >> 32);
    words[0] = (uint32_t)g0;
    for (int i = 1; i < N; ++i) {
        const uint64_t g = __atomic_load_n(&m->g[i], __ATOMIC_RELAXED);
        if ((uint32_t)(g >> 32) != t) return false;
        words[i] = (uint32_t)g;
    }
    std::memcpy(out, words, sizeof(State));
    *tag = t;
    return true;
}

// Host-side writer (tests only; the device writer is publish_mirror in cnf_mfma_dev.h)
template <class State>
static inline void cnf_mirror_write(volatile CnfMirrorT<State>* m, const State& s, uint32_t tag, const int* order = nullptr) {
    constexpr int N = CnfMirrorT<State>::kWords;
    uint32_t words[N];
    std::memcpy(words, &s, sizeof(State));
    for (int k = 0; k < N; ++k) {
        const int i = order ? order[k] : k;
        __atomic_store_n(&m->g[i], ((uint64_t)tag << 32) | words[i], __ATOMIC_RELAXED);
    }
}
