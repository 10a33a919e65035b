// Robustness driver for the host-side readers, meant to be built with -fsanitize=address,undefined (see
// tests/test_host_sanitizers.py): feeds the object-file readers, the scene cache reader, the PNG reader and the INI
// reader with the good files, with truncations of them and with random byte changes, and checks that every call
// comes back (with a result or an error) without the sanitizers firing. Test infrastructure.
//   host_sanitize <objects dir> <scratch dir> <seed.png> <seed.ini>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dirent.h>
#include <fstream>
#include <random>
#include <sstream>
#include <string>
#include <vector>

#include "config_parser.h"
#include "image_io.h"
#include "object_loader.h"
#include "scene_cache.h"
#include "scene_data.h"

using namespace rta;

static std::string ReadAll(const std::string& p) {
    std::ifstream f(p, std::ios::binary);
    std::stringstream ss;
    ss << f.rdbuf();
    return ss.str();
}
static void WriteAll(const std::string& p, const std::string& bytes) {
    std::ofstream f(p, std::ios::binary | std::ios::trunc);
    f.write(bytes.data(), (std::streamsize)bytes.size());
}

// Variants of `bytes`: cut short at several lengths, and with a few bytes replaced.
static std::vector<std::string> Variants(const std::string& bytes, std::mt19937& rng, int nRandom, bool text) {
    std::vector<std::string> out;
    for (size_t cut : {(size_t)0, (size_t)1, bytes.size() / 7, bytes.size() / 3, bytes.size() / 2, bytes.size() - std::min<size_t>(bytes.size(), 1),
                       bytes.size() - std::min<size_t>(bytes.size(), 5)})
        out.push_back(bytes.substr(0, cut));
    static const char* kText[] = {"-", "1e999", "nan", "999999999999", " ", "\n", "0", "OFF", "x", "4294967295", "-1", "."};
    for (int k = 0; k < nRandom && !bytes.empty(); ++k) {
        std::string v = bytes;
        int edits = 1 + (int)(rng() % 4);
        for (int e = 0; e < edits; ++e) {
            size_t at = rng() % v.size();
            if (text && (rng() & 1)) { const char* t = kText[rng() % (sizeof kText / sizeof *kText)]; v.replace(at, std::min<size_t>(v.size() - at, 1 + rng() % 6), t); }
            else v[at] = (char)(rng() & 0xff);
            if (v.empty()) break;
        }
        out.push_back(v);
    }
    return out;
}

int main(int argc, char** argv) {
    if (argc < 5) { fprintf(stderr, "usage: host_sanitize <objects dir> <scratch dir> <seed.png> <seed.ini>\n"); return 2; }
    const std::string objects = argv[1], scratch = argv[2], pngSeed = argv[3], iniSeed = argv[4];
    std::mt19937 rng(12345);
    std::vector<std::string> files;
    if (DIR* d = opendir(objects.c_str())) {
        while (dirent* e = readdir(d)) if (e->d_name[0] != '.') files.push_back(e->d_name);
        closedir(d);
    }
    if (files.empty()) { fprintf(stderr, "no object files in %s\n", objects.c_str()); return 2; }
    size_t calls = 0, okLoads = 0;

    // 1. object files: as they are, then damaged
    std::vector<std::string> good;
    for (const std::string& name : files) {
        LoadedObject obj;
        std::string msg;
        if (LoadObjectFile(objects + "/" + name, 0, obj, &msg) == LoadStatus::Ok) { good.push_back(objects + "/" + name); ++okLoads; }
        ++calls;
        const std::string bytes = ReadAll(objects + "/" + name);
        const bool off = IsOffPath(name);
        for (const std::string& v : Variants(bytes, rng, 120, true)) {
            const std::string p = scratch + (off ? "/variant.off" : "/variant.txt");
            WriteAll(p, v);
            LoadedObject o2;
            (void)LoadObjectFile(p, 3, o2, &msg);
            ++calls;
        }
    }
    (void)LoadObjectFile(scratch + "/does-not-exist.txt", 0, *new LoadedObject(), nullptr);

    // 2. whole scenes, one and four build threads; the scene cache and damaged blobs
    for (int threads : {1, 4}) {
        SceneData s = LoadScene(good, threads);
        if (s.TriangleCount() == 0) { fprintf(stderr, "the good files gave an empty scene\n"); return 1; }
        ++calls;
    }
    CacheResult res;
    SceneData first = LoadSceneCached(good, 2, scratch + "/cache", &res);
    SceneData again = LoadSceneCached(good, 2, scratch + "/cache", &res);
    if (res != CacheResult::kHit || again.tri12 != first.tri12 || again.node8 != first.node8) { fprintf(stderr, "scene cache: second load is not a hit with equal arrays\n"); return 1; }
    uint8_t key[32];
    if (!SceneCacheKey(good, key)) return 1;
    const std::string blobPath = scratch + "/blob.rtscene";
    if (!WriteSceneBlob(blobPath, key, first)) return 1;
    const std::string blob = ReadAll(blobPath);
    for (const std::string& v : Variants(blob, rng, 400, false)) {
        WriteAll(scratch + "/variant.rtscene", v);
        SceneData s;
        const bool accepted = ReadSceneBlob(scratch + "/variant.rtscene", key, s);
        if (accepted && v != blob && (s.tri12 != first.tri12 || s.node8 != first.node8 || s.matIdx != first.matIdx)) {
            fprintf(stderr, "scene cache: a damaged blob was accepted with different arrays\n");
            return 1;
        }
        ++calls;
    }

    // 2b. forged headers with a valid checksum: counts chosen so that the size the header implies wraps around to the file's
    // real size (offsets: nTri 40, nNode 44, nMaterials 48, nWarnings 52, warningBytes 96 in the 104-byte header)
    {
        auto forged = [&](size_t offset, uint64_t value, size_t width) {
            std::string v = blob;
            std::memcpy(&v[offset], &value, width);
            const uint64_t sum = SceneBlobChecksum(reinterpret_cast<const uint8_t*>(v.data()), v.size() - 8);
            std::memcpy(&v[v.size() - 8], &sum, 8);
            return v;
        };
        uint32_t nTri = 0, nNode = 0, nMat = 0;
        uint64_t warnBytes = 0;
        std::memcpy(&nTri, &blob[40], 4); std::memcpy(&nNode, &blob[44], 4); std::memcpy(&nMat, &blob[48], 4); std::memcpy(&warnBytes, &blob[96], 8);
        std::vector<std::string> cases;
        // one more material, paid for by "negative" warning bytes: the sum is the file size again
        cases.push_back(forged(96, warnBytes - 44, 8));
        { std::string v = cases.back(); const uint32_t m = nMat + 1; std::memcpy(&v[48], &m, 4); const uint64_t sum = SceneBlobChecksum(reinterpret_cast<const uint8_t*>(v.data()), v.size() - 8); std::memcpy(&v[v.size() - 8], &sum, 8); cases.back() = v; }
        cases.push_back(forged(96, ~0ull - 1000, 8));                   // warning bytes far beyond the file
        cases.push_back(forged(48, 0xffffffffu, 4));                     // four billion materials
        cases.push_back(forged(40, 0x01000000u, 4));                     // more triangles than indices can address
        cases.push_back(forged(44, 0xfffffff0u, 4));                     // nodes far beyond the file
        cases.push_back(forged(52, 0xffffffffu, 4));                     // more warnings than warning bytes
        for (const std::string& v : cases) {
            WriteAll(scratch + "/forged.rtscene", v);
            SceneData s;
            if (ReadSceneBlob(scratch + "/forged.rtscene", key, s)) { fprintf(stderr, "scene cache: a blob with a forged header was accepted\n"); return 1; }
            ++calls;
        }
    }

    // 3. PNG reader
    const std::string png = ReadAll(pngSeed);
    Image8 img;
    std::string err;
    if (!LoadPng(pngSeed, 3, img, &err)) { fprintf(stderr, "seed PNG does not load: %s\n", err.c_str()); return 1; }
    for (const std::string& v : Variants(png, rng, 1500, false)) {
        WriteAll(scratch + "/variant.png", v);
        Image8 i2;
        (void)LoadPng(scratch + "/variant.png", (rng() & 1) ? 3 : 4, i2, &err);
        ++calls;
    }

    // 4. INI reader (throws on malformed lines)
    const std::string ini = ReadAll(iniSeed);
    for (const std::string& v : Variants(ini, rng, 300, true)) {
        WriteAll(scratch + "/variant.ini", v);
        try {
            ConfigParser cfg(scratch + "/variant.ini");
            (void)cfg.Get<int>("Objects", "Filenames", 2);
            (void)cfg.GetAll<float>("SkyBox", "Path");
        } catch (const std::exception&) {
        }
        ++calls;
    }
    printf("host_sanitize: %zu calls, %zu of %zu object files load\n", calls, okLoads, files.size());
    return 0;
}
