// std_sort_replay / ss_heap_sort of octree_paths.h must leave (key, payload) arrays exactly as libstdc++'s
// std::sort / std::partial_sort(first, last, last) do with the same comparator, including the order of ties.
#include <algorithm>
#include <cstdio>
#include <random>
#include <vector>

#include "../../fasttrack_amd/csrc/octree_paths.h"

using ft::op::SortElem;

int main() {
    std::mt19937 rng(12345);
    int bad = 0;
    for (int trial = 0; trial < 20000; trial++) {
        const int n = (trial < 200) ? trial : 1 + (int)(rng() % 1500);
        const int distinct = 1 + (int)(rng() % (trial % 3 == 0 ? 4 : 200));  // many ties
        std::vector<SortElem> a(n), b;
        for (int i = 0; i < n; i++) a[i] = SortElem{(uint32_t)(rng() % distinct), (uint32_t)i};
        if (trial % 7 == 0) std::sort(a.begin(), a.end(), [](const SortElem &x, const SortElem &y) { return x.key < y.key; });
        if (trial % 11 == 0) std::reverse(a.begin(), a.end());
        b = a;
        const std::vector<SortElem> orig = a;
        std::sort(a.begin(), a.end(), [](const SortElem &x, const SortElem &y) { return x.key < y.key; });
        ft::op::std_sort_replay(b.data(), b.data() + n);
        for (int i = 0; i < n; i++)
            if (a[i].key != b[i].key || a[i].val != b[i].val) { bad++; break; }
        {  // the data-parallel statement of the same sort (what the device kernel runs)
            std::vector<SortElem> e = orig, tmp(n + 1);
            std::vector<uint16_t> pa(n + 1), pb(n + 1);
            ft::op::SortFrame stack[64];
            ft::op::std_sort_replay_steps(e.data(), e.data() + n, stack, pa.data(), pb.data(), tmp.data());
            for (int i = 0; i < n; i++)
                if (a[i].key != e[i].key || a[i].val != e[i].val) { bad++; break; }
        }
        // heap-sort fallback path
        std::vector<SortElem> c(n), d;
        for (int i = 0; i < n; i++) c[i] = SortElem{(uint32_t)(rng() % distinct), (uint32_t)i};
        d = c;
        std::partial_sort(c.begin(), c.end(), c.end(), [](const SortElem &x, const SortElem &y) { return x.key < y.key; });
        ft::op::ss_heap_sort(d.data(), d.data() + n);
        for (int i = 0; i < n; i++)
            if (c[i].key != d[i].key || c[i].val != d[i].val) { bad++; break; }
    }
    // an input that drives introsort into its depth limit (median-of-3 killer sequence)
    for (int n : {1024, 4096}) {
        std::vector<SortElem> a(n);
        // Musser's construction: evens ascending then odds pattern
        for (int i = 0; i < n / 2; i++) { a[i] = SortElem{(uint32_t)(i % 2 ? i + n / 2 : i / 2 * 2 + 1 + (uint32_t)0), (uint32_t)i}; }
        for (int i = n / 2; i < n; i++) a[i] = SortElem{(uint32_t)((i - n / 2) * 2), (uint32_t)i};
        std::vector<SortElem> b = a, e = a, tmp(n + 1);
        std::vector<uint16_t> pa(n + 1), pb(n + 1);
        ft::op::SortFrame stack[64];
        std::sort(a.begin(), a.end(), [](const SortElem &x, const SortElem &y) { return x.key < y.key; });
        ft::op::std_sort_replay(b.data(), b.data() + n);
        ft::op::std_sort_replay_steps(e.data(), e.data() + n, stack, pa.data(), pb.data(), tmp.data());
        for (int i = 0; i < n; i++)
            if (a[i].key != b[i].key || a[i].val != b[i].val || a[i].key != e[i].key || a[i].val != e[i].val) { bad++; break; }
    }
    std::printf("mismatches %d\n", bad);
    return bad ? 1 : 0;
}
