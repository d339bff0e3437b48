// TEST-ONLY stand-in for <hipcub/hipcub.hpp>: the one call the engine makes (DeviceRadixSort::SortPairs, the bounded cache's tail
// list) as a stable sort on the host.  Nothing in the product includes this file.
#pragma once
#include <algorithm>
#include <numeric>
#include <vector>
namespace hipcub {
struct DeviceRadixSort {
    template <class K, class V>
    static hipError_t SortPairs(void* tmp, size_t& tmp_bytes, const K* keys_in, K* keys_out, const V* vals_in, V* vals_out, int n, int begin_bit, int end_bit,
                                hipStream_t = nullptr) {
        if (!tmp) { tmp_bytes = 64; return hipSuccess; }
        const K mask = end_bit - begin_bit >= (int)(8 * sizeof(K)) ? ~(K)0 : (((K)1 << (end_bit - begin_bit)) - 1);
        std::vector<int> idx((size_t)n);
        std::iota(idx.begin(), idx.end(), 0);
        std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return ((keys_in[a] >> begin_bit) & mask) < ((keys_in[b] >> begin_bit) & mask); });
        std::vector<K> ko((size_t)n); std::vector<V> vo((size_t)n);
        for (int i = 0; i < n; ++i) { ko[i] = keys_in[idx[i]]; vo[i] = vals_in[idx[i]]; }
        std::copy(ko.begin(), ko.end(), keys_out); std::copy(vo.begin(), vo.end(), vals_out);
        return hipSuccess;
    }
};
}  // namespace hipcub
