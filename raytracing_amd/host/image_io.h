// PNG input for the renderer's textures (the reference decodes them with stb_image:
// src/main.cpp:64,96). 8-bit, non-interlaced, grey / grey+alpha / RGB / RGBA / palette.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

namespace rta {

struct Image8 {
    uint32_t width = 0, height = 0;
    int channels = 0;                 // as requested from LoadPng
    std::vector<uint8_t> pixels;      // rows top to bottom as stored in the file (no flip, like stbi_load)
};

// Decodes `path` and converts to `wantChannels` (3 or 4), the way stbi_load's req_comp does:
// grey is replicated, missing alpha becomes 255, extra alpha is dropped.
bool LoadPng(const std::string& path, int wantChannels, Image8& out, std::string* error = nullptr);

}  // namespace rta
