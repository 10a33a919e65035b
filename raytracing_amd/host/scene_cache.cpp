#include "scene_cache.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cpuid.h>
#include <immintrin.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace rta {

// ---------------------------------------------------------------------------------------------- SHA-256 (FIPS 180-4)
namespace {
const uint32_t kK[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be,
    0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa,
    0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85,
    0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3,
    0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f,
    0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
inline uint32_t Rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }

void Compress(uint32_t h[8], const uint8_t* p) {
    uint32_t w[64];
    for (int i = 0; i < 16; ++i) w[i] = ((uint32_t)p[4 * i] << 24) | ((uint32_t)p[4 * i + 1] << 16) | ((uint32_t)p[4 * i + 2] << 8) | p[4 * i + 3];
    for (int i = 16; i < 64; ++i) {
        uint32_t s0 = Rotr(w[i - 15], 7) ^ Rotr(w[i - 15], 18) ^ (w[i - 15] >> 3), s1 = Rotr(w[i - 2], 17) ^ Rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
        w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
    for (int i = 0; i < 64; ++i) {
        uint32_t t1 = hh + (Rotr(e, 6) ^ Rotr(e, 11) ^ Rotr(e, 25)) + ((e & f) ^ (~e & g)) + kK[i] + w[i];
        uint32_t t2 = (Rotr(a, 2) ^ Rotr(a, 13) ^ Rotr(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
        hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
}

// The same compression function on the CPU's SHA extensions (x86 SHA-NI: two rounds per sha256rnds2, the message
// schedule by sha256msg1/msg2), about ten times the speed of the portable code above; used when CPUID reports them.
__attribute__((target("sha,sse4.1,ssse3"))) void CompressNi(uint32_t h[8], const uint8_t* p, size_t blocks) {
    const __m128i bswap = _mm_set_epi64x(0x0c0d0e0f08090a0bULL, 0x0405060700010203ULL);
    __m128i t = _mm_loadu_si128(reinterpret_cast<const __m128i*>(h));            // DCBA
    __m128i s1 = _mm_loadu_si128(reinterpret_cast<const __m128i*>(h + 4));       // HGFE
    t = _mm_shuffle_epi32(t, 0xB1);                                              // CDAB
    s1 = _mm_shuffle_epi32(s1, 0x1B);                                            // EFGH
    __m128i s0 = _mm_alignr_epi8(t, s1, 8);                                      // ABEF
    s1 = _mm_blend_epi16(s1, t, 0xF0);                                           // CDGH
    for (; blocks; --blocks, p += 64) {
        const __m128i save0 = s0, save1 = s1;
        __m128i w[4];
        for (int g = 0; g < 16; ++g) {
            if (g < 4) {
                w[g] = _mm_shuffle_epi8(_mm_loadu_si128(reinterpret_cast<const __m128i*>(p + 16 * g)), bswap);
            } else {
                // W[4g..4g+3] = sigma1-completed( W[4g-16..] + sigma0(W[4g-15..]) + W[4g-7..4g-4] )
                __m128i x = _mm_sha256msg1_epu32(w[(g - 4) & 3], w[(g - 3) & 3]);
                x = _mm_add_epi32(x, _mm_alignr_epi8(w[(g - 1) & 3], w[(g - 2) & 3], 4));
                w[g & 3] = _mm_sha256msg2_epu32(x, w[(g - 1) & 3]);
            }
            __m128i m = _mm_add_epi32(w[g & 3], _mm_loadu_si128(reinterpret_cast<const __m128i*>(kK + 4 * g)));
            s1 = _mm_sha256rnds2_epu32(s1, s0, m);
            m = _mm_shuffle_epi32(m, 0x0E);
            s0 = _mm_sha256rnds2_epu32(s0, s1, m);
        }
        s0 = _mm_add_epi32(s0, save0);
        s1 = _mm_add_epi32(s1, save1);
    }
    t = _mm_shuffle_epi32(s0, 0x1B);                                             // FEBA
    s1 = _mm_shuffle_epi32(s1, 0xB1);                                            // DCHG
    s0 = _mm_blend_epi16(t, s1, 0xF0);                                           // DCBA
    s1 = _mm_alignr_epi8(s1, t, 8);                                              // HGFE
    _mm_storeu_si128(reinterpret_cast<__m128i*>(h), s0);
    _mm_storeu_si128(reinterpret_cast<__m128i*>(h + 4), s1);
}

bool HaveShaNi() {
    static const bool have = [] {
        if (std::getenv("RT_SHA256_PORTABLE")) return false;          // tests: force the portable code
        unsigned a = 0, b = 0, c = 0, d = 0;
        if (!__get_cpuid_count(7, 0, &a, &b, &c, &d)) return false;
        const bool sha = (b >> 29) & 1u;
        if (!__get_cpuid(1, &a, &b, &c, &d)) return false;
        return sha && ((c >> 19) & 1u) && ((c >> 9) & 1u);           // + SSE4.1, SSSE3
    }();
    return have;
}

void CompressBlocks(uint32_t h[8], const uint8_t* p, size_t blocks) {
    if (HaveShaNi()) { CompressNi(h, p, blocks); return; }
    for (; blocks; --blocks, p += 64) Compress(h, p);
}
}  // namespace

Sha256::Sha256() {
    static const uint32_t init[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    std::memcpy(h, init, sizeof h);
}

void Sha256::Update(const void* data, size_t n) {
    const uint8_t* p = static_cast<const uint8_t*>(data);
    bytes += n;
    if (fill) {
        size_t take = std::min(n, 64 - fill);
        std::memcpy(block + fill, p, take);
        fill += take; p += take; n -= take;
        if (fill < 64) return;
        CompressBlocks(h, block, 1);
        fill = 0;
    }
    if (n >= 64) { CompressBlocks(h, p, n / 64); p += n & ~(size_t)63; n &= 63; }
    if (n) { std::memcpy(block, p, n); fill = n; }
}

void Sha256::Final(uint8_t out[32]) {
    const uint64_t bits = bytes * 8;
    const uint8_t one = 0x80, zero = 0;
    Update(&one, 1);
    while (fill != 56) Update(&zero, 1);
    uint8_t len[8];
    for (int i = 0; i < 8; ++i) len[i] = (uint8_t)(bits >> (56 - 8 * i));
    Update(len, 8);
    for (int i = 0; i < 8; ++i) { out[4 * i] = h[i] >> 24; out[4 * i + 1] = h[i] >> 16; out[4 * i + 2] = h[i] >> 8; out[4 * i + 3] = h[i]; }
}

std::string HexOf(const uint8_t* p, size_t n) {
    static const char* d = "0123456789abcdef";
    std::string s;
    for (size_t i = 0; i < n; ++i) { s.push_back(d[p[i] >> 4]); s.push_back(d[p[i] & 15]); }
    return s;
}

// ------------------------------------------------------------------------------------------------------- the blob
namespace {
constexpr uint32_t kMagic = 0x43535452u;        // "RTSC"
constexpr uint32_t kFormatVersion = 1;          // bump when the loaders, the builder or this layout change their output

struct BlobHeader {
    uint32_t magic, version;
    uint8_t key[32];
    uint32_t nTri, nNode, nMaterials, nWarnings;
    int32_t maxDepth; uint32_t leaves;
    uint64_t splits, degenerateSplits, leafDepthSum;
    double buildMs;
    uint64_t warningBytes;
};
static_assert(sizeof(BlobHeader) == 104, "the blob header is part of the file format");

// A whole file, mapped read-only (nothing is copied until the caller copies it).
class MappedFile {
public:
    explicit MappedFile(const std::string& path) {
        const int fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) return;
        struct stat st;
        if (::fstat(fd, &st) == 0 && S_ISREG(st.st_mode)) {
            size_ = (size_t)st.st_size;
            ok_ = true;
            if (size_) {
                void* p = ::mmap(nullptr, size_, PROT_READ, MAP_PRIVATE, fd, 0);
                if (p == MAP_FAILED) { ok_ = false; size_ = 0; } else data_ = static_cast<const uint8_t*>(p);
            }
        }
        ::close(fd);
    }
    ~MappedFile() { if (data_) ::munmap(const_cast<uint8_t*>(data_), size_); }
    MappedFile(const MappedFile&) = delete;
    MappedFile& operator=(const MappedFile&) = delete;
    bool ok() const { return ok_; }
    const uint8_t* data() const { return data_; }
    size_t size() const { return size_; }
private:
    const uint8_t* data_ = nullptr;
    size_t size_ = 0;
    bool ok_ = false;
};

// Checksum of the blob (guards against truncated or damaged files, not against an adversary): FNV-1a steps over
// 64-bit words in four interleaved lanes, the lanes and the length folded at the end. One pass at memory speed.
uint64_t Checksum(const uint8_t* p, size_t n) {
    const uint64_t prime = 0x100000001b3ULL;
    uint64_t lane[4] = {0xcbf29ce484222325ULL, 0x84222325cbf29ce4ULL, 0x9ce484222325cbf2ULL, 0x2325cbf29ce48422ULL};
    size_t i = 0;
    for (; i + 32 <= n; i += 32) {
        uint64_t w[4];
        std::memcpy(w, p + i, 32);
        for (int k = 0; k < 4; ++k) lane[k] = (lane[k] ^ w[k]) * prime;
    }
    uint64_t h = n;
    for (; i < n; ++i) h = (h ^ p[i]) * prime;
    for (int k = 0; k < 4; ++k) h = (h ^ lane[k]) * prime;
    return h;
}
}  // namespace

bool SceneCacheKey(const std::vector<std::string>& paths, uint8_t key[32]) {
    Sha256 sha;
    const uint32_t head[3] = {kFormatVersion, (uint32_t)BvhTree::kDefaultLeafTriangles, (uint32_t)paths.size()};
    sha.Update(head, sizeof head);
    for (const std::string& p : paths) {
        MappedFile file(p);
        const bool ok = file.ok();
        // what the loader makes of a file also depends on whether its NAME says OFF (Scene.cpp:90 picks the loader by it)
        const uint64_t meta[3] = {ok ? 1u : 0u, ok ? (uint64_t)file.size() : 0u, (uint64_t)(p.find(".off") != std::string::npos || p.find(".OFF") != std::string::npos)};
        sha.Update(meta, sizeof meta);
        if (ok && file.size()) sha.Update(file.data(), file.size());
    }
    sha.Final(key);
    return true;
}

bool WriteSceneBlob(const std::string& path, const uint8_t key[32], const SceneData& s) {
    BlobHeader h{};
    h.magic = kMagic; h.version = kFormatVersion;
    std::memcpy(h.key, key, 32);
    h.nTri = s.TriangleCount(); h.nNode = s.NodeCount(); h.nMaterials = (uint32_t)s.materials.size(); h.nWarnings = (uint32_t)s.warnings.size();
    h.maxDepth = s.stats.maxDepth; h.leaves = s.stats.leaves; h.splits = s.stats.splits; h.degenerateSplits = s.stats.degenerateSplits;
    h.leafDepthSum = s.stats.leafDepthSum; h.buildMs = s.stats.buildMs;
    std::string warn;
    for (const std::string& w : s.warnings) { warn += w; warn.push_back('\0'); }
    h.warningBytes = warn.size();
    // assembled in memory first: the checksum runs over the file's bytes as a whole
    std::vector<uint8_t> blob;
    blob.reserve(sizeof h + s.tri12.size() * 4 + s.matIdx.size() * 4 + s.node8.size() * 4 + s.materials.size() * sizeof(Material) + warn.size() + 8);
    auto put = [&](const void* p, size_t n) { const uint8_t* b = static_cast<const uint8_t*>(p); if (n) blob.insert(blob.end(), b, b + n); };
    static_assert(sizeof(Material) == 11 * sizeof(float), "materials are stored as 11 floats");
    put(&h, sizeof h); put(s.tri12.data(), s.tri12.size() * 4); put(s.matIdx.data(), s.matIdx.size() * 4);
    put(s.node8.data(), s.node8.size() * 4); put(s.materials.data(), s.materials.size() * sizeof(Material)); put(warn.data(), warn.size());
    const uint64_t tail = Checksum(blob.data(), blob.size());
    put(&tail, 8);
    const std::string tmp = path + ".part" + std::to_string((long)::getpid());
    FILE* f = std::fopen(tmp.c_str(), "wb");
    if (!f) return false;
    bool ok = std::fwrite(blob.data(), 1, blob.size(), f) == blob.size();
    ok = (std::fclose(f) == 0) && ok;
    if (ok) ok = std::rename(tmp.c_str(), path.c_str()) == 0;
    if (!ok) std::remove(tmp.c_str());
    return ok;
}

uint64_t SceneBlobChecksum(const uint8_t* bytes, size_t n) { return Checksum(bytes, n); }

bool ReadSceneBlob(const std::string& path, const uint8_t key[32], SceneData& s) {
    MappedFile file(path);
    if (!file.ok() || file.size() < sizeof(BlobHeader) + 8) return false;
    const uint8_t* b = file.data();
    BlobHeader h;
    std::memcpy(&h, b, sizeof h);
    if (h.magic != kMagic || h.version != kFormatVersion || std::memcmp(h.key, key, 32) != 0) return false;
    // Every count comes from the file: each term is bounded by the file's size before it is added, so that the sum
    // cannot wrap around to the right value while a single term reaches far beyond the mapping.
    const uint64_t size = file.size();
    if (h.nTri >= (1u << 24) || h.warningBytes > size || (uint64_t)h.nMaterials * sizeof(Material) > size ||
        (uint64_t)h.nNode * 32 > size || h.nWarnings > h.warningBytes) return false;
    uint64_t need = sizeof h + 8;
    if (__builtin_add_overflow(need, (uint64_t)h.nTri * 52, &need) || __builtin_add_overflow(need, (uint64_t)h.nNode * 32, &need) ||
        __builtin_add_overflow(need, (uint64_t)h.nMaterials * sizeof(Material), &need) || __builtin_add_overflow(need, h.warningBytes, &need)) return false;
    if (need != size) return false;
    uint64_t tail;
    std::memcpy(&tail, b + file.size() - 8, 8);
    if (tail != Checksum(b, file.size() - 8)) return false;
    const uint8_t* p = b + sizeof h;
    s = SceneData();
    auto floats = [&](std::vector<float>& v, size_t n) { const float* f = reinterpret_cast<const float*>(p); v.assign(f, f + n); p += n * 4; };
    floats(s.tri12, (size_t)h.nTri * 12);
    { const int32_t* m = reinterpret_cast<const int32_t*>(p); s.matIdx.assign(m, m + h.nTri); p += (size_t)h.nTri * 4; }
    floats(s.node8, (size_t)h.nNode * 8);
    s.materials.resize(h.nMaterials);
    if (h.nMaterials) std::memcpy(s.materials.data(), p, (size_t)h.nMaterials * sizeof(Material));
    p += (size_t)h.nMaterials * sizeof(Material);
    const char* w = reinterpret_cast<const char*>(p);
    const char* end = w + h.warningBytes;
    for (uint32_t i = 0; i < h.nWarnings && w < end; ++i) { size_t n = strnlen(w, (size_t)(end - w)); s.warnings.emplace_back(w, n); w += n + 1; }
    if (s.warnings.size() != h.nWarnings) return false;
    s.stats.maxDepth = h.maxDepth; s.stats.leaves = h.leaves; s.stats.splits = h.splits; s.stats.degenerateSplits = h.degenerateSplits;
    s.stats.leafDepthSum = h.leafDepthSum; s.stats.buildMs = h.buildMs;
    return true;
}

SceneData LoadSceneCached(const std::vector<std::string>& paths, int buildThreads, const std::string& cacheDir, CacheResult* result) {
    auto report = [&](CacheResult r) { if (result) *result = r; };
    if (cacheDir.empty()) { report(CacheResult::kDisabled); return LoadScene(paths, buildThreads); }
    uint8_t key[32];
    SceneCacheKey(paths, key);
    const std::string blob = cacheDir + "/" + HexOf(key, 32) + ".rtscene";
    SceneData s;
    if (ReadSceneBlob(blob, key, s)) { report(CacheResult::kHit); return s; }
    s = LoadScene(paths, buildThreads);
    ::mkdir(cacheDir.c_str(), 0777);
    report(WriteSceneBlob(blob, key, s) ? CacheResult::kMissWritten : CacheResult::kMissNotWritten);
    return s;
}

}  // namespace rta
