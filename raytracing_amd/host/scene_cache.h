// Binary cache of a loaded scene (the flat arrays Scene::LoadObjects would produce, src/Scene.cpp:87-127), so that
// a start-up after the first one neither parses the text files (Include/ObjectLoader.h:54-156) nor rebuilds the
// tree (src/BvhTree.cpp:129-151).
//
// Key: SHA-256 over a format version, the leaf size, and for every input file, in order, whether it could be
// read, its length and its bytes. Anything that changes the result changes the key: edited, reordered, added or
// missing files. The blob is <cacheDir>/<key as hex>.rtscene; it repeats its key and ends with a checksum of
// everything before it, and a blob that is short, damaged, of another version or carries another key is ignored
// (the scene is then loaded the slow way and the blob rewritten).
#pragma once

#include <string>
#include <vector>

#include "scene_data.h"

namespace rta {

struct Sha256 {
    uint32_t h[8];
    uint64_t bytes = 0;
    uint8_t block[64];
    size_t fill = 0;
    Sha256();
    void Update(const void* data, size_t n);
    void Final(uint8_t out[32]);
};
std::string HexOf(const uint8_t* p, size_t n);

enum class CacheResult { kDisabled, kHit, kMissWritten, kMissNotWritten };

// LoadScene through the cache in `cacheDir` (created if missing; empty = no cache).
SceneData LoadSceneCached(const std::vector<std::string>& objectPaths, int buildThreads, const std::string& cacheDir, CacheResult* result = nullptr);

// The pieces, exposed for tests.
bool SceneCacheKey(const std::vector<std::string>& objectPaths, uint8_t key[32]);
bool WriteSceneBlob(const std::string& path, const uint8_t key[32], const SceneData& scene);
bool ReadSceneBlob(const std::string& path, const uint8_t key[32], SceneData& scene);
// The checksum a blob ends in (over everything before it), so that a test can build blobs with a forged header.
uint64_t SceneBlobChecksum(const uint8_t* bytes, size_t n);

}  // namespace rta
