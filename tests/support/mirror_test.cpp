// CPU test of the host reader of the state mirror (continuousnf.jl_amd/csrc/cnf_mirror.h) against a writer thread
// that behaves as badly as the protocol allows: granules stored in a random order, no fences, no waits, the next
// launch's stores beginning at any time.  Every snapshot the reader accepts must be one launch's state, whole.
#include "../../continuousnf.jl_amd/csrc/cnf_mirror.h"

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <thread>

struct State { float t, dt, q; int cur, done, naccept, nreject; float h, a, r, e, d0, d1, x; int i0, i1, i2, i3, i4; };   // 19 words, like StepState

static State make(uint32_t tag) {
    State s;
    uint32_t* w = reinterpret_cast<uint32_t*>(&s);
    for (int i = 0; i < (int)(sizeof(State) / 4); ++i) w[i] = tag * 2654435761u + (uint32_t)i * 40503u;
    return s;
}
static bool whole(const State& s, uint32_t tag) {
    const State e = make(tag);
    return std::memcmp(&s, &e, sizeof s) == 0;
}

int main(int argc, char** argv) {
    const uint32_t n_writes = argc > 1 ? (uint32_t)atoi(argv[1]) : 300000u;
    static CnfMirrorT<State> m;
    std::memset(&m, 0, sizeof m);
    std::atomic<bool> stop{false};
    std::thread writer([&] {
        std::mt19937 rng(12345);
        constexpr int N = CnfMirrorT<State>::kWords;
        int order[N];
        for (int i = 0; i < N; ++i) order[i] = i;
        for (uint32_t tag = 1; tag <= n_writes; ++tag) {
            for (int i = N - 1; i > 0; --i) std::swap(order[i], order[rng() % (i + 1)]);
            cnf_mirror_write(&m, make(tag), tag, order);
            if ((rng() & 7) == 0) std::this_thread::yield();
        }
        stop = true;
    });
    uint32_t last = 0, accepted = 0, rejected = 0;
    long bad = 0;
    while (!stop.load()) {
        State s;
        uint32_t tag;
        if (cnf_mirror_read(&m, &s, &tag)) {
            if (tag != 0) {
                if (!whole(s, tag)) ++bad;
                if ((int32_t)(tag - last) < 0) ++bad;          // tags never go backwards
                last = tag;
                ++accepted;
            }
        } else {
            ++rejected;
        }
    }
    writer.join();
    State s;
    uint32_t tag = 0;
    const bool final_ok = cnf_mirror_read(&m, &s, &tag) && tag == n_writes && whole(s, tag);
    std::printf("accepted %u rejected %u bad %ld final %d\n", accepted, rejected, bad, (int)final_ok);
    return (bad == 0 && final_ok && accepted > 0) ? 0 : 1;
}
