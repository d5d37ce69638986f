// Calls the REAL replay reader of the reference (include/replay.hpp, which needs nothing but the
// standard library) compiled from where it lies -- see Makefile.  Test infrastructure: used by
// tests/tools/make_golden_replay.py to produce tests/golden/replay_cases.json and by the pin test where
// /root/reference is mounted.  Contains no reference code, only the call.
#include "replay.hpp"
#include <cstdint>

extern "C" int32_t ref_replay_load(const char* path, float* out /* 6 per tick */, int32_t max_ticks)
{
    const std::list<ReplayElements> ticks = ReplayElements::loadFromFile(path);   // replay.hpp:18
    int32_t n = 0;
    for (const ReplayElements& e : ticks) {
        if (n < max_ticks) {
            float* o = out + 6 * n;
            o[0] = e.timestamp; o[1] = e.x; o[2] = e.y; o[3] = e.z; o[4] = e.view_x; o[5] = e.view_y;
        }
        ++n;
    }
    return n;
}
