// Drives work through the REAL thread pool of the reference (lib/swarm/swarm.hpp, standard library only), compiled from
// where it lies -- see Makefile (oracle/_ref/libref_swarm.so).  Test / measurement infrastructure: bench.py's cpu_baseline
// leg times the oracle's per-area frame job on the reference's own swarm, with the reference's decomposition (16 workers,
// worker t renders area (t % 4, t / 4), main.cpp:90-92,139-154).  Contains no reference code, only the calls
// main.cpp:90,139,154 makes: Swarm(thread_count), execute(lambda), waitExecutionDone().
#include "swarm/swarm.hpp"

#include <atomic>
#include <chrono>
#include <cstdint>
#include <thread>

extern "C" void* ref_swarm_create(uint32_t thread_count) { return new swrm::Swarm(thread_count); }   // main.cpp:90-91
extern "C" void ref_swarm_destroy(void* s) { delete static_cast<swrm::Swarm*>(s); }

// One fork / join of `job(worker_id, worker_count, user)` over all workers.  Swarm::execute silently returns an empty
// group (no work done) when its workers have not re-registered yet (swarm.hpp:221-223): the number of
// jobs that actually ran is counted and the call repeated until all of them did.  Returns the number of execute()
// calls that were needed (1 = first try), 0 if the work never ran within max_tries.
extern "C" uint32_t ref_swarm_run(void* s, uint32_t thread_count, void (*job)(uint32_t, uint32_t, void*), void* user, uint32_t max_tries)
{
    swrm::Swarm* swarm = static_cast<swrm::Swarm*>(s);
    for (uint32_t attempt = 1; attempt <= max_tries; ++attempt) {
        std::atomic<uint32_t> ran(0u);
        swrm::WorkGroup group = swarm->execute([&](uint32_t id, uint32_t n) { ++ran; job(id, n, user); });   // main.cpp:139
        group.waitExecutionDone();                                                                            // main.cpp:154
        if (ran.load() == thread_count) return attempt;
        std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
    return 0u;
}
