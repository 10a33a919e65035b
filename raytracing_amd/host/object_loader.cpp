#include "object_loader.h"

#include <algorithm>
#include <cerrno>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <limits>

namespace rta {
namespace {

// Whitespace-separated token reader over an in-memory file. Like a std::istream it latches a
// failure: after the first token that does not parse, every later read fails too.
class Tokens {
public:
    explicit Tokens(std::string text) : buf_(std::move(text)), p_(buf_.c_str()) {}
    bool failed() const { return failed_; }

    bool Word(std::string& out) {
        if (failed_) return false;
        Skip();
        if (!*p_) { failed_ = true; return false; }
        const char* b = p_;
        while (*p_ && !IsSpace(*p_)) ++p_;
        out.assign(b, p_);
        return true;
    }
    // Accepts what `istream >> float` accepts: an optional sign, decimal digits with an optional point, an optional
    // decimal exponent. strtof alone would also take "nan", "inf" and hexadecimal floats, which the reference's
    // stream extraction (Include/ObjectLoader.h:137-139) rejects or cuts short.
    bool Float(float& out) {
        if (failed_) return false;
        Skip();
        const char* q = p_;
        if (*q == '+' || *q == '-') ++q;
        const char* digits = q;
        while (*q >= '0' && *q <= '9') ++q;
        if (*q == '.') { ++q; while (*q >= '0' && *q <= '9') ++q; }
        if (q == digits || (q == digits + 1 && *digits == '.')) { failed_ = true; return false; }
        if (*q == 'e' || *q == 'E') {
            const char* e = q + 1;
            if (*e == '+' || *e == '-') ++e;
            if (*e >= '0' && *e <= '9') { while (*e >= '0' && *e <= '9') ++e; q = e; }
        }
        char* end = nullptr;
        float v = std::strtof(p_, &end);
        if (end != q) {
            // strtof read on past the decimal syntax (a hexadecimal float): convert the decimal prefix alone
            std::string prefix(p_, q);
            v = std::strtof(prefix.c_str(), &end);
            if (end == prefix.c_str()) { failed_ = true; return false; }
        }
        p_ = q;
        if (std::isinf(v)) {
            // a literal too large for a float: the stream extraction the reference uses stores the largest finite value
            // and fails, which ends the object (Include/ObjectLoader.h:137-139)
            out = v > 0 ? std::numeric_limits<float>::max() : -std::numeric_limits<float>::max();
            failed_ = true;
            return false;
        }
        out = v;
        return true;
    }
    bool Int(int& out) {
        if (failed_) return false;
        Skip();
        char* end = nullptr;
        errno = 0;
        long v = std::strtol(p_, &end, 10);
        if (end == p_) { failed_ = true; out = 0; return false; }
        p_ = end;
        // a number that does not fit an int fails the read and leaves the nearest limit, as `istream >> int` does
        if (errno == ERANGE || v > INT_MAX || v < INT_MIN) { failed_ = true; out = v < 0 ? INT_MIN : INT_MAX; return false; }
        out = (int)v;
        return true;
    }
    // An upper bound on how many more items of at least `minBytes` characters each the text can hold: what a count
    // read from the file is cut down to before memory is set aside for it.
    size_t RoomFor(long count, size_t minBytes) const {
        if (count <= 0) return 0;
        const size_t left = buf_.size() - (size_t)(p_ - buf_.c_str());
        return std::min<size_t>((size_t)count, left / minBytes + 1);
    }

private:
    static bool IsSpace(char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\f' || c == '\v'; }
    void Skip() { while (*p_ && IsSpace(*p_)) ++p_; }
    std::string buf_;
    const char* p_;
    bool failed_ = false;
};

struct Placement {
    Vec3 position;         // added after scaling
    float scale = 1.0f;
    bool swapYZ = false;   // `format xzy`
    Vec3 Apply(Vec3 v) const {
        if (swapYZ) std::swap(v.y, v.z);
        return v * scale + position;
    }
};

bool ReadFile(const std::string& path, std::string& out) {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    std::fseek(f, 0, SEEK_END);
    long n = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    out.resize(n > 0 ? (size_t)n : 0);
    size_t got = n > 0 ? std::fread(&out[0], 1, (size_t)n, f) : 0;
    std::fclose(f);
    out.resize(got);
    return true;
}

// Header keywords, then the material (Include/ObjectLoader.h:68-131).
bool ReadHeaderAndMaterial(Tokens& in, Placement& place, Material& material, std::string* msg) {
    std::string word, kind;
    while (in.Word(word)) {
        if (word == "position") {
            in.Float(place.position.x); in.Float(place.position.z); in.Float(place.position.y);   // file order is x z y
        } else if (word == "scale") {
            in.Float(place.scale);
        } else if (word == "format") {
            std::string f;
            if (in.Word(f)) {
                if (f == "xzy") place.swapYZ = true;
                else place.swapYZ = false;     // "xyz", and any invalid value falls back to xyz
            }
        } else { kind = word; break; }
    }
    if (kind == "default-material") { material = Material::Lambertian({0.75f, 0.75f, 0.75f}); return true; }
    Vec3 c;
    if (!(in.Float(c.x) && in.Float(c.y) && in.Float(c.z))) { if (msg) *msg = "unable to read material color"; return false; }
    bool ok = true;
    if (kind == "specular") { float r; ok = in.Float(r); if (ok) material = Material::Specular(c, r); }
    else if (kind == "lambertian") { material = Material::Lambertian(c); }
    else if (kind == "metallic") { float r, m; ok = in.Float(r) && in.Float(m); if (ok) material = Material::Metallic(c, r, m); }
    else if (kind == "transparent") { float t, n; ok = in.Float(t) && in.Float(n); if (ok) material = Material::Transparent(c, t, n); }
    else if (kind == "lightsource") { float l; ok = in.Float(l); if (ok) material = Material::LightSource(c, l); }
    else { if (msg) *msg = "unknown material type: " + kind; return false; }
    if (!ok && msg) *msg = "unable to read material format";
    return ok;
}

// n, then n triangles of nine floats (Include/ObjectLoader.h:133-156).
bool ReadRawTriangles(Tokens& in, const Placement& place, int slot, std::vector<Triangle>& tris, std::string* msg) {
    int n = 0;
    in.Int(n);
    tris.reserve(in.RoomFor(n, 18));          // nine numbers and their separators
    for (int i = 0; i < n; ++i) {
        Vec3 a, b, c;
        bool ok = in.Float(a.x) && in.Float(a.y) && in.Float(a.z) && in.Float(b.x) && in.Float(b.y) && in.Float(b.z) &&
                  in.Float(c.x) && in.Float(c.y) && in.Float(c.z);
        if (!ok) { if (msg) *msg = "failed to read triangle " + std::to_string(i + 1); return false; }
        tris.emplace_back(place.Apply(a), place.Apply(b), place.Apply(c), slot);
    }
    return true;
}

// OFF body (Include/ObjectLoader.h:164-229).
bool ReadOff(Tokens& in, const Placement& place, int slot, std::vector<Triangle>& tris, std::string* msg) {
    std::string magic;
    in.Word(magic);   // a wrong magic word is reported by the reference but not fatal
    int nv, nf, ne;
    if (!(in.Int(nv) && in.Int(nf) && in.Int(ne))) { if (msg) *msg = "failed to read OFF counts"; return false; }
    std::vector<Vec3> verts;
    verts.reserve(in.RoomFor(nv, 6));
    for (int i = 0; i < nv; ++i) {
        Vec3 v;
        if (!(in.Float(v.x) && in.Float(v.y) && in.Float(v.z))) { if (msg) *msg = "failed to read vertex " + std::to_string(i + 1); return false; }
        verts.push_back(place.Apply(v));
    }
    if (nf != 0) {
        tris.reserve(in.RoomFor(nf, 8));
        for (int i = 0; i < nf; ++i) {
            int k, a, b, c;
            if (!(in.Int(k) && in.Int(a) && in.Int(b) && in.Int(c))) { if (msg) *msg = "failed to read face " + std::to_string(i + 1); return false; }
            if (a < 0 || b < 0 || c < 0 || a >= nv || b >= nv || c >= nv) {   // the reference indexes unchecked
                if (msg) *msg = "face " + std::to_string(i + 1) + " references a vertex out of range";
                return false;
            }
            tris.emplace_back(verts[a], verts[b], verts[c], slot);
        }
        return true;
    }
    // Point cloud: one axis-aligned cube of side 0.02 per vertex, 12 triangles, in this corner order.
    const float h = 0.02f * 0.5f;
    static const int kFaces[12][3] = {{0, 1, 2}, {0, 2, 3}, {4, 5, 6}, {4, 6, 7}, {0, 1, 5}, {0, 5, 4},
                                      {3, 2, 6}, {3, 6, 7}, {0, 3, 7}, {0, 7, 4}, {1, 2, 6}, {1, 6, 5}};
    tris.reserve(verts.size() * 12);
    for (const Vec3& ctr : verts) {
        const Vec3 corner[8] = {ctr + Vec3(-h, -h, -h), ctr + Vec3(h, -h, -h), ctr + Vec3(h, h, -h), ctr + Vec3(-h, h, -h),
                                ctr + Vec3(-h, -h, h),  ctr + Vec3(h, -h, h),  ctr + Vec3(h, h, h),  ctr + Vec3(-h, h, h)};
        for (const auto& f : kFaces) tris.emplace_back(corner[f[0]], corner[f[1]], corner[f[2]], slot);
    }
    return true;
}

}  // namespace

bool IsOffPath(const std::string& path) {
    return path.find(".off") != std::string::npos || path.find(".OFF") != std::string::npos;
}

LoadStatus LoadObjectFile(const std::string& path, int materialSlot, LoadedObject& out, std::string* message) {
    std::string text;
    if (!ReadFile(path, text)) { if (message) *message = "unable to read from: " + path; return LoadStatus::CannotOpen; }
    Tokens in(std::move(text));
    Placement place;
    out.triangles.clear();
    if (!ReadHeaderAndMaterial(in, place, out.material, message)) return LoadStatus::BadMaterial;
    bool ok = IsOffPath(path) ? ReadOff(in, place, materialSlot, out.triangles, message)
                              : ReadRawTriangles(in, place, materialSlot, out.triangles, message);
    if (!ok) { out.triangles.clear(); return LoadStatus::BadGeometry; }
    return LoadStatus::Ok;
}

}  // namespace rta
