// Readers for the renderer's two scene-file formats (reference: Include/ObjectLoader.h,
// object_settings.md):
//   header keywords in any order  `position x z y` | `scale s` | `format xyz|xzy`
//   one material line             default-material | lambertian r g b | specular r g b rough |
//                                 metallic r g b rough metal | transparent r g b transp ior |
//                                 lightsource r g b luminance
//   geometry                      raw list:  n, then n x 9 floats          (any file name)
//                                 OFF     :  OFF nv nf ne, nv vertices, nf faces `k a b c`
//                                            (file name contains .off / .OFF); nf == 0 turns each
//                                            vertex into a 0.02-wide cube of 12 triangles
#pragma once

#include <optional>
#include <string>
#include <vector>

#include "geometry.h"

namespace rta {

struct LoadedObject {
    Material material;
    std::vector<Triangle> triangles;
};

enum class LoadStatus { Ok, CannotOpen, BadMaterial, BadGeometry };

// Parses one object file. `materialSlot` is stored in every triangle (the renderer's material
// table index of this object). On BadGeometry `out.material` is still valid, as in the
// reference, where the material is registered before the geometry is read (src/Scene.cpp:101-111).
LoadStatus LoadObjectFile(const std::string& path, int materialSlot, LoadedObject& out, std::string* message = nullptr);

// True when the path selects the OFF reader (src/Scene.cpp:91).
bool IsOffPath(const std::string& path);

}  // namespace rta
