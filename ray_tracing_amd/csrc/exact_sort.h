// std::sort's exact result, computed in parallel.
//
// The reference's BVH builders sort triangle ranges with std::sort (c5:505-507, 579-581, 431-436); std::sort is not stable,
// so where centroid coordinates tie, the triangle order — and with it every later split — is whatever this library's
// introsort happens to produce (the permutation is pinned by this container's libstdc++).  A different sort
// would give a different, equally valid tree, but not the reference build's tree.  This header therefore does not
// re-implement the algorithm: it drives libstdc++'s own building blocks (std::__unguarded_partition_pivot,
// std::__introsort_loop, std::__partial_sort, std::__final_insertion_sort — bits/stl_algo.h) in the order std::__sort
// does, and only changes WHO runs them:
//   * introsort phase: after a partition step the two sides are disjoint ranges whose processing never looks outside
//     them, so the right side becomes an OpenMP task while the loop continues on the left side, exactly the recursion /
//     iteration split of std::__introsort_loop; ranges below kSeqRange are finished by std::__introsort_loop itself;
//   * final insertion sort: a partition cut is a position where everything on the left is <= everything on the right
//     (the pivot stays at the front of the left side), so the unguarded linear insertion of an element never moves it —
//     or looks — past a cut to its left.  The recorded cuts split the array into chunks that are insertion-sorted
//     independently (even chunks, then odd ones, so the sentinel element left of a chunk is never being written while it
//     is read); the first chunk takes std::__final_insertion_sort's guarded head.
// Every comparison and every move is one the sequential std::sort performs, on the same operands: the resulting
// permutation is identical, ties included (tests/exact_sort_test.cpp compares the two on tie-heavy, sorted, organ-pipe
// and depth-limit-exhausting inputs).  Requires a strict weak order (no NaN keys): callers check and fall back to
// std::sort otherwise.  Call from inside an OpenMP parallel region (tasks); outside one it degrades to sequential order.
#pragma once
#include <algorithm>
#include <mutex>
#include <vector>

#ifndef __GLIBCXX__
#error "exact_sort.h drives libstdc++'s introsort internals; the result is pinned to that library"
#endif

namespace rta {

constexpr long kSeqRange = 1 << 15;      // introsort ranges up to this long are finished by the library's sequential loop
constexpr long kParSortMin = 1 << 17;    // shorter arrays: plain std::sort
constexpr long kInsertChunkMin = 1 << 14;  // final insertion sort: cuts closer than this to the previous boundary are not used

namespace detail {

template <class T, class IterCmp>
struct ParSort {
    IterCmp comp;
    std::mutex mu;
    std::vector<T*> cuts;

    void loop(T* first, T* last, long depth_limit) {
        while (last - first > (long)std::_S_threshold) {
            if (last - first <= kSeqRange) {
                std::__introsort_loop(first, last, depth_limit, comp);
                return;
            }
            if (depth_limit == 0) {
                std::__partial_sort(first, last, last, comp);
                return;
            }
            --depth_limit;
            T* const cut = std::__unguarded_partition_pivot(first, last, comp);
            {
                std::lock_guard<std::mutex> g(mu);
                cuts.push_back(cut);
            }
            T* const rlast = last;
            const long rdepth = depth_limit;
#pragma omp task default(shared) firstprivate(cut, rlast, rdepth)
            loop(cut, rlast, rdepth);
            last = cut;
        }
    }

    void run(T* first, T* last, long depth_limit) {
#pragma omp taskgroup
        loop(first, last, depth_limit);
        // chunk boundaries for the final insertion sort: cuts at least kInsertChunkMin apart
        std::sort(cuts.begin(), cuts.end());
        std::vector<T*> b;
        b.push_back(first);
        for (T* c : cuts)
            if (c - b.back() >= kInsertChunkMin && last - c >= kInsertChunkMin) b.push_back(c);
        b.push_back(last);
        const int n_chunks = (int)b.size() - 1;
        for (int phase = 0; phase < 2; ++phase) {
#pragma omp taskgroup
            for (int j = phase; j < n_chunks; j += 2) {
#pragma omp task default(shared) firstprivate(j)
                {
                    if (j == 0)
                        std::__final_insertion_sort(b[0], b[1], comp);
                    else
                        std::__unguarded_insertion_sort(b[j], b[j + 1], comp);
                }
            }
        }
    }
};

}  // namespace detail

// [first, last) sorted exactly as std::sort(first, last, cmp) leaves it.  `depth_limit` < 0: std::sort's own (2 * floor(log2 n)).
template <class T, class Cmp>
void exact_sort(T* first, T* last, Cmp cmp, long depth_limit = -1) {
    if (last - first < kParSortMin && depth_limit < 0) {
        std::sort(first, last, cmp);
        return;
    }
    if (first == last) return;
    auto ic = __gnu_cxx::__ops::__iter_comp_iter(cmp);
    detail::ParSort<T, decltype(ic)> ps{ic, {}, {}};
    ps.run(first, last, depth_limit < 0 ? (long)std::__lg(last - first) * 2 : depth_limit);
}

// the sequential counterpart with an explicit depth limit (test hook: std::sort never takes one)
template <class T, class Cmp>
void std_sort_with_depth(T* first, T* last, Cmp cmp, long depth_limit) {
    if (first == last) return;
    auto ic = __gnu_cxx::__ops::__iter_comp_iter(cmp);
    std::__introsort_loop(first, last, depth_limit, ic);
    std::__final_insertion_sort(first, last, ic);
}

}  // namespace rta
