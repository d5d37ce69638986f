// AddressSanitizer / UBSan driver for the host LSVO builder (csrc/vrc_builder.cpp): CPU build only.
#include "vrc.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
int main()
{
    const uint32_t d = 6, S = 64;
    std::vector<int32_t> h(S * S);
    for (uint32_t i = 0; i < S * S; i++) h[i] = (int32_t)((i * 2654435761u) >> 26) - 10;
    vrc_lnode* n;
    uint64_t c;
    int rc = vrc_build_terrain_lsvo(h.data(), d, &n, &c);
    std::printf("terrain rc %d nodes %lu\n", rc, (unsigned long)c);
    vrc_free_host(n);
    std::vector<uint8_t> v(S * S * S);
    for (size_t i = 0; i < v.size(); i++) v[i] = ((uint32_t)(i * 2654435761u) >> 29) == 0;
    rc = vrc_build_volume_lsvo(v.data(), d, &n, &c);
    std::printf("volume rc %d nodes %lu\n", rc, (unsigned long)c);
    vrc_free_host(n);
    std::vector<uint8_t> e(8 * 8 * 8, 0);
    rc = vrc_build_volume_lsvo(e.data(), 3, &n, &c);
    std::printf("empty rc %d nodes %lu\n", rc, (unsigned long)c);
    vrc_free_host(n);
    return (rc == 0 && c == 1) ? 0 : 1;
}
