// Stand-in for hipCUB under the host emulation of the tests: the one routine the library uses, as a stable host sort.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <numeric>
#include <vector>

namespace hipcub {
struct DeviceRadixSort {
    template <class K, class V>
    static hipError_t SortPairs(void* temp, size_t& tempBytes, const K* keysIn, K* keysOut, const V* valuesIn, V* valuesOut, int n, int beginBit = 0,
                                int endBit = (int)sizeof(K) * 8, hipStream_t = nullptr) {
        if (!temp) { tempBytes = 256; return hipSuccess; }
        const K mask = (endBit - beginBit >= (int)sizeof(K) * 8) ? ~K(0) : (K)(((K(1) << (endBit - beginBit)) - 1) << beginBit);
        std::vector<int> idx((size_t)n);
        std::iota(idx.begin(), idx.end(), 0);
        std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return (keysIn[a] & mask) < (keysIn[b] & mask); });
        for (int i = 0; i < n; ++i) { keysOut[i] = keysIn[idx[i]]; valuesOut[i] = valuesIn[idx[i]]; }
        return hipSuccess;
    }
};
}  // namespace hipcub
