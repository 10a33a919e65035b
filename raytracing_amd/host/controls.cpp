#include "controls.h"

#include <cmath>

namespace rta {

namespace {
constexpr double kPiOn180 = 0.01745329251994329576923690768489;   // src/Camera.cpp:9

// Rodrigues rotation, src/Camera.cpp:66-71. The reference's unqualified cos(a)/sin(a) resolve to the
// double-precision functions (only <cmath> is included), and the results are narrowed to float.
Vec3 RotateAroundAxis(const Vec3& v, const Vec3& k, float a) {
    float cosA = (float)std::cos((double)a), sinA = (float)std::sin((double)a);
    return (v * cosA + cross(k, v) * sinA) + k * dot(k, v) * (1 - cosA);
}
}  // namespace

Camera::Camera() : position_(0, 0, 0), facing_(1, 0, 0) {
    fov_ = (float)(120 * kPiOn180);
    viewportDistance_ = 1.0f;
    viewportWidth_ = (float)(2.0 * viewportDistance_ * std::tan(fov_ / 2.0));
}

void Camera::SetFovDegrees(float degrees) {
    fov_ = (float)((double)degrees * kPiOn180);
    viewportWidth_ = (float)(2.0 * viewportDistance_ * std::tan(fov_ / 2.0));
}

void Camera::OnEvent(const KeyEvent& e) {
    if (e.key < 0 || e.key >= kKeyCount) return;
    if (e.action == kPress) keys_[e.key] = true;
    else if (e.action == kRelease) keys_[e.key] = false;
}

bool Camera::Tick(float cameraSpeed, float rotationSpeed) {
    const Vec3 p0 = position_, f0 = facing_;
    const float fov0 = fov_;
    const Vec3 right = normalized(cross(facing_, Vec3(0.0f, 0.0f, 1.0f)));
    const Vec3 up = normalized(cross(right, facing_));
    if (keys_[kKeyLeftShift]) { cameraSpeed *= 2; rotationSpeed *= 2; }
    else if (keys_[kKeyRightShift]) { cameraSpeed /= 2; rotationSpeed /= 2; }
    if (keys_[kKeyW]) position_ = position_ + facing_ * cameraSpeed;
    if (keys_[kKeyS]) position_ = position_ - facing_ * cameraSpeed;
    if (keys_[kKeyA]) position_ = position_ - right * cameraSpeed;
    if (keys_[kKeyD]) position_ = position_ + right * cameraSpeed;
    if (keys_[kKeySpace]) position_ = position_ + Vec3(0, 0, 1) * (cameraSpeed / 2);
    if (keys_[kKeyLeftControl]) position_ = position_ - Vec3(0, 0, 1) * (cameraSpeed / 2);
    if (keys_[kKeyUp]) facing_ = RotateAroundAxis(facing_, right, rotationSpeed);
    if (keys_[kKeyDown]) facing_ = RotateAroundAxis(facing_, right, -rotationSpeed);
    if (keys_[kKeyLeft]) facing_ = RotateAroundAxis(facing_, up, rotationSpeed);
    if (keys_[kKeyRight]) facing_ = RotateAroundAxis(facing_, up, -rotationSpeed);
    SetFovDegrees(keys_[kKeyC] ? 90.0f : 120.0f);
    return !(f0 == facing_) || !(p0 == position_) || !(fov0 == fov_);
}

std::optional<uint32_t> BounceLimitFromKey(int key) {
    switch (key) {
        case kKeyGrave: return 0u;   // BVH heat map
        case kKey1: return 1u;       // normals
        case kKey2: return 2u;
        case kKey3: return 4u;
        case kKey4: return 8u;
        case kKey5: return 16u;
        case kKey6: return 32u;
        default: return std::nullopt;
    }
}

GlibcRand::GlibcRand(uint32_t seed) {
    r_.resize(34);
    r_[0] = (int32_t)(seed ? seed : 1);
    for (int i = 1; i < 31; ++i) {
        int64_t v = (16807LL * r_[i - 1]) % 2147483647LL;
        if (v < 0) v += 2147483647LL;
        r_[i] = (int32_t)v;
    }
    for (int i = 31; i < 34; ++i) r_[i] = r_[i - 31];
    i_ = 34;
    for (int k = 0; k < 310; ++k) (void)Next();   // glibc discards the first 310 outputs
}

int GlibcRand::Next() {
    uint32_t v = (uint32_t)r_[i_ - 31] + (uint32_t)r_[i_ - 3];
    r_.push_back((int32_t)v);
    ++i_;
    if (r_.size() > 4096) { r_.erase(r_.begin(), r_.end() - 64); i_ = r_.size(); }
    return (int)(v >> 1);
}

void GlibcRand::NextSeedTriple(float out[3]) {
    const float rmax = (float)2147483647;   // RAND_MAX promoted to float
    out[2] = (float)Next() / rmax;
    out[1] = (float)Next() / rmax;
    out[0] = (float)Next() / rmax;
}

}  // namespace rta
