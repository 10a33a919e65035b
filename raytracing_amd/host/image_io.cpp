#include "image_io.h"

#include <zlib.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace rta {
namespace {

uint32_t Be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

int Paeth(int a, int b, int c) {
    int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

bool Fail(std::string* e, const char* msg) { if (e) *e = msg; return false; }

constexpr uint32_t kMaxPngSide = 16384;

}  // namespace

bool LoadPng(const std::string& path, int want, Image8& out, std::string* error) {
    if (want != 3 && want != 4) return Fail(error, "wantChannels must be 3 or 4");
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return Fail(error, "cannot open file");
    std::vector<uint8_t> file;
    uint8_t buf[65536];
    size_t n;
    while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) file.insert(file.end(), buf, buf + n);
    std::fclose(f);
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (file.size() < 8 || std::memcmp(file.data(), sig, 8) != 0) return Fail(error, "not a PNG file");

    uint32_t w = 0, h = 0;
    int depth = 0, colour = 0, interlace = 0;
    std::vector<uint8_t> idat, palette, trns;
    for (size_t pos = 8; pos + 12 <= file.size();) {
        uint32_t len = Be32(&file[pos]);
        const uint8_t* type = &file[pos + 4];
        const uint8_t* data = &file[pos + 8];
        if (pos + 12 + len > file.size()) return Fail(error, "truncated chunk");
        if (!std::memcmp(type, "IHDR", 4) && len >= 13) { w = Be32(data); h = Be32(data + 4); depth = data[8]; colour = data[9]; interlace = data[12]; }
        else if (!std::memcmp(type, "PLTE", 4)) palette.assign(data, data + len);
        else if (!std::memcmp(type, "tRNS", 4)) trns.assign(data, data + len);
        else if (!std::memcmp(type, "IDAT", 4)) idat.insert(idat.end(), data, data + len);
        else if (!std::memcmp(type, "IEND", 4)) break;
        pos += 12 + (size_t)len;
    }
    const bool packed = depth < 8 && (colour == 0 || colour == 3) && (depth == 1 || depth == 2 || depth == 4);
    if (!w || !h || (depth != 8 && !packed) || interlace != 0) return Fail(error, "unsupported PNG (need up to 8 bits per sample, non-interlaced)");
    // the header is untrusted: bound the allocations below (the renderer takes textures up to 16384 texels a side)
    if (w > kMaxPngSide || h > kMaxPngSide) return Fail(error, "image larger than 16384 x 16384");
    if (idat.empty()) return Fail(error, "no image data");
    int src = colour == 0 ? 1 : colour == 2 ? 3 : colour == 3 ? 1 : colour == 4 ? 2 : colour == 6 ? 4 : 0;
    if (!src) return Fail(error, "unsupported colour type");
    const size_t stride = packed ? ((size_t)w * depth + 7) / 8 : (size_t)w * src;
    std::vector<uint8_t> raw((stride + 1) * h);
    uLongf rawLen = (uLongf)raw.size();
    if (uncompress(raw.data(), &rawLen, idat.data(), (uLong)idat.size()) != Z_OK || rawLen != raw.size()) return Fail(error, "bad image data");
    std::vector<uint8_t> img(stride * h);
    for (uint32_t y = 0; y < h; ++y) {
        const uint8_t* in = &raw[(stride + 1) * y];
        uint8_t* cur = &img[stride * y];
        const uint8_t* up = y ? &img[stride * (y - 1)] : nullptr;
        const int filter = in[0];
        for (size_t i = 0; i < stride; ++i) {
            int a = i >= (size_t)src ? cur[i - src] : 0, b = up ? up[i] : 0, c = (up && i >= (size_t)src) ? up[i - src] : 0;
            int x = in[1 + i];
            switch (filter) {
                case 0: break;
                case 1: x += a; break;
                case 2: x += b; break;
                case 3: x += (a + b) / 2; break;
                case 4: x += Paeth(a, b, c); break;
                default: return Fail(error, "bad filter type");
            }
            cur[i] = (uint8_t)x;
        }
    }
    if (packed) {                                            // 1/2/4-bit samples -> one byte each
        std::vector<uint8_t> wide((size_t)w * h);
        const int mask = (1 << depth) - 1, scale = colour == 0 ? 255 / mask : 1;
        for (uint32_t y = 0; y < h; ++y)
            for (uint32_t x = 0; x < w; ++x) {
                size_t bit = (size_t)x * depth;
                int v = (img[stride * y + bit / 8] >> (8 - depth - (int)(bit % 8))) & mask;
                wide[(size_t)y * w + x] = (uint8_t)(v * scale);
            }
        img.swap(wide);
    }
    out.width = w; out.height = h; out.channels = want;
    out.pixels.resize((size_t)w * h * want);
    for (size_t p = 0; p < (size_t)w * h; ++p) {
        uint8_t r, g, b, a = 255;
        const uint8_t* s = &img[p * src];
        if (colour == 0) { r = g = b = s[0]; }
        else if (colour == 4) { r = g = b = s[0]; a = s[1]; }
        else if (colour == 3) {
            size_t k = s[0];
            if (k * 3 + 2 >= palette.size()) return Fail(error, "palette index out of range");
            r = palette[k * 3]; g = palette[k * 3 + 1]; b = palette[k * 3 + 2];
            if (k < trns.size()) a = trns[k];
        } else { r = s[0]; g = s[1]; b = s[2]; if (colour == 6) a = s[3]; }
        uint8_t* d = &out.pixels[p * want];
        d[0] = r; d[1] = g; d[2] = b;
        if (want == 4) d[3] = a;
    }
    return true;
}

}  // namespace rta
