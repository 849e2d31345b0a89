// kvc_stl_emul.h — the libstdc++ algorithms torch-CPU topk runs (std::partial_sort; std::nth_element + std::sort),
// restated move for move on (key << 32 | index) elements so that ties resolve exactly as they do there.
// Sources restated: GCC 11 bits/stl_heap.h (__push_heap, __adjust_heap, __make_heap, __sort_heap) and bits/stl_algo.h
// (__heap_select, __move_median_to_first, __unguarded_partition(_pivot), __introselect, __insertion_sort,
// __unguarded_linear_insert, __introsort_loop, __final_insertion_sort).  Compiles for the device (hipcc) and for the
// host (g++, used by tests/test_stl_emul.py to check it against the real library on tie-heavy inputs).
//
// LICENCE NOTE.  This file follows the control flow of the GNU ISO C++ Library (libstdc++, Copyright (C) 2001-2021 Free Software
// Foundation, Inc.; bits/stl_heap.h and bits/stl_algo.h also carry Copyright (c) 1994 Hewlett-Packard Company and Copyright (c)
// 1996, 1997 Silicon Graphics Computer Systems, Inc. notices: "Permission to use, copy, modify, distribute and sell this software
// and its documentation for any purpose is hereby granted without fee, provided that the above copyright notice appear in all copies
// and that both that copyright notice and this permission notice appear in supporting documentation.  [The companies make] no
// representations about the suitability of this software for any purpose.  It is provided "as is" without express or implied
// warranty.").  libstdc++ is distributed under the GNU General Public License version 3, or (at your option) any later version,
// with the GCC Runtime Library Exception version 3.1.  The algorithms are re-typed here, not copied, because the ORDER OF MOVES of
// these routines is the specification of torch-CPU's tie order; whoever redistributes this file should treat it as a derived work
// of those headers under the same terms (GPL-3.0-or-later WITH GCC-exception-3.1) — see NOTICE at the repository root.  Nothing
// else in the repository derives from libstdc++ sources.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define KVC_HD __device__
#else
#define KVC_HD
#endif

namespace kvc {

typedef unsigned long long u64;

#if defined(__HIPCC__)
// every lane of the wave runs the same program: make loaded values wave-uniform so branches are scalar
KVC_HD inline u64 uni(u64 v) {
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return ((u64)hi << 32) | lo;
}
KVC_HD inline int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
#else
inline u64 uni(u64 v) { return v; }
inline int uni(int v) { return v; }
#endif

// comp(a, b): a sorts before b  <=>  value(a) > value(b)   (keys are order-preserving; every NaN has the one key above +inf's,
// kvc_common.h Key<>: the key comparison IS torch's NaN-first comparator)
KVC_HD inline bool comp(u64 a, u64 b) { return (uint32_t)(a >> 32) > (uint32_t)(b >> 32); }

struct Arr {                 // random-access view of the (key<<32 | index) array, LDS or global
    u64* p;
    KVC_HD inline u64 get(int i) const { return uni(p[i]); }
    KVC_HD inline void set(int i, u64 v) const { p[i] = v; }
    KVC_HD inline void swap(int i, int j) const { const u64 a = get(i), b = get(j); p[i] = b; p[j] = a; }
};

// (A register-resident variant of this view — heap slots on lanes, v_readlane/compare-select access — was measured
// in round 1 and is SLOWER than LDS for a lone wave: the scalar program is issue-latency bound, not LDS bound.)

// ---- libstdc++ bits/stl_heap.h ----------------------------------------------------------------------------
template <class AT>
KVC_HD inline void push_heap_(AT& A, int first, int hole, int top, u64 value) {
    int parent = (hole - 1) / 2;
    while (hole > top && comp(A.get(first + parent), value)) {
        A.set(first + hole, A.get(first + parent));
        hole = parent;
        parent = (hole - 1) / 2;
    }
    A.set(first + hole, value);
}
template <class AT>
KVC_HD inline void adjust_heap_(AT& A, int first, int hole, int len, u64 value) {
    const int top = hole;
    int second = hole;
    while (second < (len - 1) / 2) {
        second = 2 * (second + 1);
        if (comp(A.get(first + second), A.get(first + second - 1))) second--;
        A.set(first + hole, A.get(first + second));
        hole = second;
    }
    if ((len & 1) == 0 && second == (len - 2) / 2) {
        second = 2 * (second + 1);
        A.set(first + hole, A.get(first + second - 1));
        hole = second - 1;
    }
    push_heap_(A, first, hole, top, value);
}
template <class AT>
KVC_HD inline void make_heap_(AT& A, int first, int last) {
    const int len = last - first;
    if (len < 2) return;
    int parent = (len - 2) / 2;
    while (true) {
        const u64 value = A.get(first + parent);
        adjust_heap_(A, first, parent, len, value);
        if (parent == 0) return;
        parent--;
    }
}
template <class AT>
KVC_HD inline void sort_heap_(AT& A, int first, int last) {
    while (last - first > 1) {
        --last;
        const u64 value = A.get(last);
        A.set(last, A.get(first));
        adjust_heap_(A, first, 0, last - first, value);
    }
}
// heap_select over an array that is fully materialised in A
template <class AT>
KVC_HD inline void heap_select_(AT& A, int first, int middle, int last) {
    make_heap_(A, first, middle);
    for (int i = middle; i < last; ++i) {
        const u64 vi = A.get(i);
        if (comp(vi, A.get(first))) {
            A.set(i, A.get(first));
            adjust_heap_(A, first, 0, middle - first, vi);
        }
    }
}

// ---- libstdc++ bits/stl_algo.h ----------------------------------------------------------------------------
KVC_HD inline int lg_(int n) { return 31 - __builtin_clz(n); }

template <class AT>
KVC_HD inline void move_median_to_first_(AT& A, int result, int a, int b, int c) {
    const u64 va = A.get(a), vb = A.get(b), vc = A.get(c);
    if (comp(va, vb)) {
        if (comp(vb, vc)) A.swap(result, b);
        else if (comp(va, vc)) A.swap(result, c);
        else A.swap(result, a);
    } else if (comp(va, vc)) A.swap(result, a);
    else if (comp(vb, vc)) A.swap(result, c);
    else A.swap(result, b);
}
template <class AT>
KVC_HD inline int unguarded_partition_(AT& A, int first, int last, int pivot) {
    const u64 pv = A.get(pivot);
    while (true) {
        while (comp(A.get(first), pv)) ++first;
        --last;
        while (comp(pv, A.get(last))) --last;
        if (!(first < last)) return first;
        A.swap(first, last);
        ++first;
    }
}
template <class AT>
KVC_HD inline int unguarded_partition_pivot_(AT& A, int first, int last) {
    const int mid = first + (last - first) / 2;
    move_median_to_first_(A, first, first + 1, mid, last - 1);
    return unguarded_partition_(A, first + 1, last, first);
}
template <class AT>
KVC_HD inline void unguarded_linear_insert_(AT& A, int last) {
    const u64 val = A.get(last);
    int next = last - 1;
    while (comp(val, A.get(next))) {
        A.set(last, A.get(next));
        last = next;
        --next;
    }
    A.set(last, val);
}
template <class AT>
KVC_HD inline void insertion_sort_(AT& A, int first, int last) {
    if (first == last) return;
    for (int i = first + 1; i != last; ++i) {
        const u64 vi = A.get(i);
        if (comp(vi, A.get(first))) {
            for (int j = i; j > first; --j) A.set(j, A.get(j - 1));     // move_backward(first, i, i + 1)
            A.set(first, vi);
        } else {
            unguarded_linear_insert_(A, i);
        }
    }
}
template <class AT>
KVC_HD inline void introselect_(AT& A, int first, int nth, int last, int depth_limit) {
    while (last - first > 3) {
        if (depth_limit == 0) {
            heap_select_(A, first, nth + 1, last);
            A.swap(first, nth);
            return;
        }
        --depth_limit;
        const int cut = unguarded_partition_pivot_(A, first, last);
        if (cut <= nth) first = cut; else last = cut;
    }
    insertion_sort_(A, first, last);
}
// std::sort = introsort loop (explicit stack instead of recursion) + final insertion sort
template <class AT>
KVC_HD inline void sort_(AT& A, int first, int last, int* stack /*LDS, 3 ints per frame, >= 64 frames*/) {
    if (first == last) return;
    int sp = 0;
    int f = first, l = last, d = lg_(last - first) * 2;
    while (true) {
        while (l - f > 16) {
            if (d == 0) {                       // __partial_sort(first, last, last): heap sort of the range
                heap_select_(A, f, l, l);
                sort_heap_(A, f, l);
                break;
            }
            --d;
            const int cut = unguarded_partition_pivot_(A, f, l);
            // recurse on [cut, l) first (libstdc++ order), then continue with [f, cut): ranges are disjoint, so
            // deferring the right part on a stack yields the same final array.
            stack[3 * sp] = cut; stack[3 * sp + 1] = l; stack[3 * sp + 2] = d; ++sp;
            l = cut;
        }
        if (sp == 0) break;
        --sp;
        f = uni(stack[3 * sp]); l = uni(stack[3 * sp + 1]); d = uni(stack[3 * sp + 2]);
    }
    if (last - first > 16) {
        insertion_sort_(A, first, first + 16);
        for (int i = first + 16; i != last; ++i) unguarded_linear_insert_(A, i);
    } else {
        insertion_sort_(A, first, last);
    }
}

// The first `want` elements of std::sort(first, last) as indices in out[0..want) (AdaKV / HeadKV keep a prefix of a head's full
// descending sort).  Scalar statement of what WaveSel::sort_prefix_to (kvc_select_exact.hip) does lane-parallel: the
// introsort loop descends only into ranges that start before first + want — a range further right is never mixed with
// anything to its left again — and every element below first + want + 16 is placed by its stable-insertion count
// (insertion sort is stable and the loop leaves ordered runs of at most 16).  Checked against std::sort by tests/stl_emul_host.cpp.
template <class AT>
KVC_HD inline void sort_prefix_(AT& A, int first, int last, int want, int* stack, int64_t* out) {
    if (first == last || want <= 0) return;
    int sp = 0;
    int f = first, l = last, d = lg_(last - first) * 2;
    while (true) {
        while (l - f > 16) {
            if (d == 0) { heap_select_(A, f, l, l); sort_heap_(A, f, l); break; }
            --d;
            const int cut = unguarded_partition_pivot_(A, f, l);
            if (cut < first + want) { stack[3 * sp] = cut; stack[3 * sp + 1] = l; stack[3 * sp + 2] = d; ++sp; }
            l = cut;
        }
        if (sp == 0) break;
        --sp;
        f = uni(stack[3 * sp]); l = uni(stack[3 * sp + 1]); d = uni(stack[3 * sp + 2]);
    }
    const int lim = first + want + 16 < last ? first + want + 16 : last;
    for (int i = first; i < lim; ++i) {
        const u64 v = A.get(i);
        int pos = i;
        for (int dlt = 1; dlt < 16; ++dlt) {
            const int jb = i - dlt, ja = i + dlt;
            if (jb >= first && comp(v, A.get(jb))) --pos;               // key_jb < key_i: it ends up behind i
            if (ja < last && comp(A.get(ja), v)) ++pos;                 // key_ja > key_i: it ends up in front of i
        }
        if (pos - first < want) out[pos - first] = (int64_t)(v & 0xffffffffull);
    }
}

}  // namespace kvc
