// The reference's interactive control surface, without a window system: key events are injected
// (from a script, a test, or any UI) and interpreted with the reference's rules.
//   Camera              <- Include/Camera.h, src/Camera.cpp        (WASD / space / ctrl / arrows / shift / C)
//   BounceLimitFromKey  <- src/BounceLimitManager.cpp:14-48        (` 1 2 3 4 5 6)
//   GlibcRand           <- the unseeded std::rand() stream that feeds u_RandSeed (src/Scene.cpp:31)
#pragma once

#include <cstdint>
#include <optional>
#include <vector>

#include "geometry.h"

namespace rta {

// GLFW key codes and actions used by the reference (GLFW/glfw3.h values).
enum Key : int {
    kKeySpace = 32, kKey1 = 49, kKey2 = 50, kKey3 = 51, kKey4 = 52, kKey5 = 53, kKey6 = 54, kKeyA = 65, kKeyB = 66, kKeyC = 67, kKeyD = 68,
    kKeyP = 80, kKeyR = 82, kKeyS = 83, kKeyW = 87, kKeyGrave = 96, kKeyEscape = 256, kKeyRight = 262, kKeyLeft = 263, kKeyDown = 264,
    kKeyUp = 265, kKeyLeftShift = 340, kKeyLeftControl = 341, kKeyRightShift = 344, kKeyCount = 350
};
enum KeyAction : int { kRelease = 0, kPress = 1, kRepeat = 2 };

struct KeyEvent { int key; int action; };

class Camera {
public:
    Camera();                                     // position (0,0,0), facing (1,0,0), fov 120 degrees, viewport distance 1

    void OnEvent(const KeyEvent& e);              // src/Camera.cpp:137-143
    // One frame of movement (src/Camera.cpp:78-127). Returns true when position, facing or fov
    // changed, i.e. when the reference re-uploads the camera and restarts the accumulation.
    bool Tick(float cameraSpeed = 0.2f, float rotationSpeed = 0.02f);

    const Vec3& Position() const { return position_; }
    const Vec3& Facing() const { return facing_; }
    float Fov() const { return fov_; }
    float ViewportWidth() const { return viewportWidth_; }
    float ViewportDistance() const { return viewportDistance_; }
    void SetPosition(const Vec3& p) { position_ = p; }
    void SetFacing(const Vec3& f) { facing_ = normalized(f); }   // src/Camera.cpp:46-48
    void SetFovDegrees(float degrees);                           // src/Camera.cpp:73-76

private:
    Vec3 position_, facing_;
    float fov_ = 0.0f, viewportWidth_ = 0.0f, viewportDistance_ = 1.0f;
    bool keys_[kKeyCount] = {false};
};

// Bounce limit selected by a key press, if the key is one of ` 1 2 3 4 5 6 -> 0 1 2 4 8 16 32.
std::optional<uint32_t> BounceLimitFromKey(int key);

// glibc's default random() generator (TYPE_3, seed 1), which is what an unseeded std::rand()
// returns on Linux. Re-stated so that seed streams do not depend on the C library in use.
class GlibcRand {
public:
    explicit GlibcRand(uint32_t seed = 1);
    int Next();                                   // 0 .. 2^31-1
    // The three values of one Scene::Tick: g++ evaluates the call's arguments right to left, so the
    // first draw lands in .z. Each is float(rand()) / RAND_MAX.
    void NextSeedTriple(float out[3]);

private:
    std::vector<int32_t> r_;
    size_t i_ = 0;
};

}  // namespace rta
