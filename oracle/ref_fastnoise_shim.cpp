// Calls the REAL vendored FastNoise of the reference (compiled from
// /root/reference/lib/fastnoise/FastNoise.cpp where it lies -- see Makefile)
// exactly the way the reference's scene generator does (src/main.cpp:61-69).
// Test infrastructure: used by tests/tools/make_golden.py to produce the terrain
// height fixture.  Contains no reference code, only the calls.
#include "FastNoise.h"
#include <cstdint>

extern "C" void ref_terrain_heights(int32_t size, int32_t* height /* [x*size+z] */)
{
    FastNoise myNoise;                                   // main.cpp:61 (seed 1337 default)
    myNoise.SetNoiseType(FastNoise::SimplexFractal);     // main.cpp:62
    for (uint32_t x = 0; x < (uint32_t)size; x++)
        for (uint32_t z = 0; z < (uint32_t)size; z++)
            height[(int64_t)x * size + z] =
                int32_t(64.0f * myNoise.GetNoise(float(0.75f * x), float(0.75f * z)) + 32); // main.cpp:69
}
