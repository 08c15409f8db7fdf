// exact_sort (csrc/exact_sort.h) must leave every array exactly as std::sort does — ties included.
// g++ -O2 -std=c++17 -fopenmp tests/exact_sort_test.cpp -o /tmp/exact_sort_test && /tmp/exact_sort_test
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <random>
#include <vector>

#include "../ray_tracing_amd/csrc/exact_sort.h"

struct Rec {
    float cx, cy, cz;
    int idx;
};
static bool by_x(const Rec& a, const Rec& b) { return a.cx < b.cx; }
static bool by_y(const Rec& a, const Rec& b) { return a.cy < b.cy; }

static std::vector<Rec> make(int kind, size_t n, uint32_t seed) {
    std::mt19937 g(seed);
    std::vector<Rec> v(n);
    for (size_t i = 0; i < n; ++i) {
        float k = 0;
        switch (kind) {
            case 0: k = (float)(g() % 1000003u) * 1e-3f; break;                   // many distinct keys, some ties
            case 1: k = (float)(g() % 17u); break;                                // 17 distinct keys: almost all comparisons tie
            case 2: k = (float)i; break;                                          // sorted
            case 3: k = (float)(n - i); break;                                    // reversed
            case 4: k = (float)(i < n / 2 ? i : n - i); break;                    // organ pipe
            case 5: k = 1.0f; break;                                              // all equal
            case 6: k = (float)((i * 2654435761u) % 4096u) + ((i & 1) ? 0.0f : -0.0f); break;  // ties, signed zeros
            case 7: k = (float)(i / 8); break;                                    // sorted runs of ties
            case 8: k = ldexpf((float)(g() & 0xffffff), -20) + 9.0f; break;       // fp32 collisions around 9..25, like centroids
            default: k = (float)(g() & 255u) + ((g() & 3u) ? 0.0f : 1e-3f * (float)(g() % 5u)); break;
        }
        v[i] = Rec{k, (float)(g() % 5u), 0.0f, (int)i};
    }
    return v;
}

static long fails = 0, cases = 0;

static void check(int kind, size_t n, uint32_t seed, long depth, bool use_y) {
    const std::vector<Rec> src = make(kind, n, seed);
    std::vector<Rec> a = src, b = src;
    auto cmp = use_y ? by_y : by_x;
    if (depth < 0)
        std::sort(a.begin(), a.end(), cmp);
    else
        rta::std_sort_with_depth(a.data(), a.data() + n, cmp, depth);
#pragma omp parallel
#pragma omp single
    rta::exact_sort(b.data(), b.data() + n, cmp, depth);
    ++cases;
    if (n && memcmp(a.data(), b.data(), n * sizeof(Rec)) != 0) {
        ++fails;
        printf("MISMATCH kind %d n %zu seed %u depth %ld\n", kind, n, seed, depth);
    }
}

int main() {
    const size_t sizes[] = {0, 1, 15, 16, 17, 1000, (size_t)rta::kSeqRange, (size_t)rta::kSeqRange + 1, (size_t)rta::kParSortMin - 1,
                            (size_t)rta::kParSortMin, 300001, 1u << 20, 2000003};
    for (int kind = 0; kind < 10; ++kind)
        for (size_t n : sizes) {
            check(kind, n, 1234u + (uint32_t)kind, -1, false);
            if (n >= 1000) check(kind, n, 99u + (uint32_t)kind, -1, true);  // second key: 5 distinct values
        }
    // explicit depth limits: the heap-sort branch of the introsort loop, at the top and below parallel splits
    for (int kind = 0; kind < 10; ++kind)
        for (long depth : {0L, 1L, 2L, 3L, 6L, 12L})
            for (size_t n : {(size_t)1000, (size_t)140000, (size_t)400000}) check(kind, n, 77u + (uint32_t)kind, depth, false);
    // sorting again what is already sorted by another key's sort (the builder's sequence: x, y, z, then the best axis)
    {
        std::vector<Rec> a = make(8, 2000003, 5), b;
        for (size_t i = 0; i < a.size(); ++i) a[i].cy = (float)((i * 40503u) % 977u);
        b = a;
        for (int round = 0; round < 4; ++round) {
            auto cmp = (round & 1) ? by_y : by_x;
            std::sort(a.begin(), a.end(), cmp);
#pragma omp parallel
#pragma omp single
            rta::exact_sort(b.data(), b.data() + b.size(), cmp);
            ++cases;
            if (memcmp(a.data(), b.data(), a.size() * sizeof(Rec)) != 0) ++fails, printf("MISMATCH in the sort sequence, round %d\n", round);
        }
    }
    printf("%ld cases, %ld mismatches\n", cases, fails);
    return fails ? 1 : 0;
}
