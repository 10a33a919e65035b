// Host-side data model of a scene: vectors, triangles, materials.
//
// Mirrors the reference's POD types so that flattened buffers come out bit-identical:
//   Vec3      <- Include/Math3D.h:26-130   (fp32 component arithmetic, no fused operations)
//   Triangle  <- Include/Tri.h:8-32        (three corners + cached bounds + centroid + material slot)
//   Material  <- Include/Materials.h:7-49  (eleven scalars, five presets)
// Build with -ffp-contract=off so that a*b+c is never fused.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

namespace rta {

struct Vec3 {
    float x = 0.0f, y = 0.0f, z = 0.0f;
    Vec3() = default;
    Vec3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
    float operator[](int axis) const { return axis == 0 ? x : axis == 1 ? y : z; }
    float& at(int axis) { return axis == 0 ? x : axis == 1 ? y : z; }
};

inline Vec3 operator+(const Vec3& a, const Vec3& b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline Vec3 operator-(const Vec3& a, const Vec3& b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Vec3 operator*(const Vec3& a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline Vec3 operator/(const Vec3& a, float s) { return {a.x / s, a.y / s, a.z / s}; }
inline bool operator==(const Vec3& a, const Vec3& b) { return a.x == b.x && a.y == b.y && a.z == b.z; }
inline float dot(const Vec3& a, const Vec3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline Vec3 cross(const Vec3& a, const Vec3& b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
// Math3D.h:50-57: each component divided by the length (not multiplied by a reciprocal).
// (A double-precision sqrt of a float sum, narrowed back to float, equals the float sqrt.)
inline Vec3 normalized(const Vec3& v) {
    float mag = std::sqrt(v.x * v.x + v.y * v.y + v.z * v.z);
    return {v.x / mag, v.y / mag, v.z / mag};
}

struct Triangle {
    Vec3 p0, p1, p2;
    Vec3 hi, lo;      // component-wise max / min of the corners
    Vec3 centroid;    // (p0 + p1 + p2) / 3, in that order
    int32_t material = 0;

    Triangle() = default;
    Triangle(const Vec3& a, const Vec3& b, const Vec3& c, int32_t mat) : p0(a), p1(b), p2(c), material(mat) {
        hi = {std::max(a.x, std::max(b.x, c.x)), std::max(a.y, std::max(b.y, c.y)), std::max(a.z, std::max(b.z, c.z))};
        lo = {std::min(a.x, std::min(b.x, c.x)), std::min(a.y, std::min(b.y, c.y)), std::min(a.z, std::min(b.z, c.z))};
        centroid = ((a + b) + c) / 3.0f;
    }
};

// Field order is the order the renderer's material table is uploaded in.
struct Material {
    Vec3 colour;
    Vec3 specularColour{1.0f, 1.0f, 1.0f};
    float roughness = 1.0f;
    float metallic = 0.0f;
    float transparency = 0.0f;   // for a light source this slot carries the luminance
    float refractionIndex = 0.0f;
    float isLight = 0.0f;

    static Material Lambertian(const Vec3& c) { Material m; m.colour = c; return m; }
    static Material Specular(const Vec3& c, float roughness) { Material m; m.colour = c; m.roughness = roughness; return m; }
    static Material Metallic(const Vec3& c, float roughness, float metallic) {
        Material m; m.colour = c; m.roughness = roughness; m.metallic = metallic; return m;
    }
    static Material Transparent(const Vec3& c, float transparency, float ior) {
        Material m; m.colour = c; m.roughness = 0.0f; m.metallic = 0.5f; m.transparency = transparency; m.refractionIndex = ior;
        return m;
    }
    static Material LightSource(const Vec3& c, float luminance) {
        Material m; m.colour = c; m.roughness = 0.0f; m.metallic = 0.0f; m.transparency = luminance; m.isLight = 1.0f;
        return m;
    }
};

}  // namespace rta
