// Compile / link check of the drop-in class (round-2 verdict item 6): HipVolumetric used exactly the way the reference uses a
// Volumetric -- through a `const Volumetric&`, as Camera::getClosestPoint does (include/camera_controller.hpp:56-60:
// `volume.castRay(position * scale + glm::vec3(1.0f), camera_vec, 0.0f, 0.0f)`) -- against the reference's OWN
// include/volumetric.hpp and include/cell.hpp.  GLM is absent from this image: <glm/glm.hpp> resolves to the labelled stand-in
// tests/standin_glm (not GLM; pins nothing).  Built and run by tests/test_host_volumetric.py where /root/reference is mounted.
#define VRC_WITH_REFERENCE_HEADERS
#include "../../cpuvoxelraycaster_amd/host/hip_raycaster.hpp"

#include <cstdio>
#include <type_traits>

static_assert(std::is_base_of<Volumetric, HipVolumetric>::value, "HipVolumetric must be a Volumetric");
static_assert(!std::is_abstract<HipVolumetric>::value, "HipVolumetric must override every pure virtual of Volumetric");

// what Camera::getClosestPoint does with its argument (camera_controller.hpp:56-60)
static HitPoint closest_point(const Volumetric& volume, const glm::vec3& position, const glm::vec3& camera_vec)
{
    constexpr float scale = 1.0f / 512.0f;
    return volume.castRay(position * scale + glm::vec3(1.0f), camera_vec, 0.0f, 0.0f);
}

int main()
{
    // a one-node tree: root with child 0 a leaf
    vrc_lnode root;
    root.color = 0; root.child_mask = 1; root.leaf_mask = 1; root.pad = 0; root.child_offset = 1;
    try {
        HipVolumetric volume(&root, 1, 9, 0);
        Volumetric& as_base = volume;
        as_base.setCell(Cell::Solid, Cell::Grass, 0u, 0u, 0u);
        const HitPoint h = closest_point(volume, glm::vec3(256.0f, 200.0f, 256.0f), glm::vec3(0.0f, 0.0f, 1.0f));
        std::printf("castRay through Volumetric&: cell=%p distance=%g complexity=%u\n", (const void*)h.cell, h.distance, h.complexity);
    } catch (const std::exception& e) {
        std::printf("exception: %s\n", e.what());      // without a HIP device: the scene cannot be created (no CPU fallback)
    }
    return 0;
}
