// STAND-IN, NOT GLM.  A minimal substitute for <glm/glm.hpp>, written for ONE purpose: to let a compiler type-check
// `HipVolumetric : public Volumetric` (cpuvoxelraycaster_amd/host/hip_raycaster.hpp) against the reference's own
// include/volumetric.hpp and include/cell.hpp in a container that has no GLM (tests/test_host_volumetric.py).
// It pins NOTHING: no numerical result of the reference is derived from this file, the oracle does not use it, and a build of
// the reference made with it would not count as the reference (SURVEY.md App. D describes what a real harness would need).
// Only what volumetric.hpp:1-61 touches exists here: vec2, vec3 (three-argument constructor from arithmetic values,
// scalar / vec3, abs).
#pragma once
#include <cmath>
#include <cstdint>

namespace glm {

struct vec2 {
    float x, y;
    vec2() : x(0.0f), y(0.0f) {}
    vec2(float x_, float y_) : x(x_), y(y_) {}
};

struct vec3 {
    float x, y, z;
    vec3() : x(0.0f), y(0.0f), z(0.0f) {}
    explicit vec3(float s) : x(s), y(s), z(s) {}
    template <class A, class B, class C>
    vec3(A x_, B y_, C z_) : x((float)x_), y((float)y_), z((float)z_) {}
};

inline vec3 operator/(float s, const vec3& v) { return vec3(s / v.x, s / v.y, s / v.z); }
inline vec3 operator*(const vec3& v, float s) { return vec3(v.x * s, v.y * s, v.z * s); }
inline vec3 operator+(const vec3& a, const vec3& b) { return vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline vec3 abs(const vec3& v) { return vec3(std::fabs(v.x), std::fabs(v.y), std::fabs(v.z)); }

}  // namespace glm
