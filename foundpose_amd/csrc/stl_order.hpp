// Tie-for-tie reproduction of `torch.topk(x, k, largest=True, sorted=True)` on a 1-D CPU tensor.
//
// The reference selects templates with torch.topk(cos_sims, 5) (utils/template_util.py:172) and best buddies with
// torch.topk(-cycle_dists, k) (utils/corresp_util.py:61) on CPU tensors.  ATen's CPU kernel copies the row into
// (value, index) pairs and runs libstdc++'s std::partial_sort (when k*64 <= n) or std::nth_element + std::sort of
// the first k-1 elements, with a strict-weak "greater" comparator on the VALUE only.  Among equal values the
// resulting order is therefore whatever those algorithms' element moves produce -- deterministic, but not by
// index.  Cycle distances are heavily tied (many exact zeros, multiples of the 14-px grid), so reproducing the
// reference's correspondence ORDER (which feeds RANSAC sampling in the unchanged PnP tail) needs the same moves.
//
// This header restates those algorithms (introselect / introsort with median-of-3 + unguarded partition, threshold
// 3 / 16, heap fallback; heap select + sort_heap) from their published descriptions, as host/device code operating
// on an array in place.  tests/test_stl_order.py checks it against the real std:: algorithms on tied inputs.
#pragma once

#if defined(__HIPCC__)
#define FP_HD __host__ __device__ inline
#else
#define FP_HD inline
#endif

namespace stl_order {

struct Elem {
  float v;
  int idx;
};

// torch's comparator for largest=True: NaN sorts first, otherwise by value; the index never participates.
FP_HD bool gt(const Elem& x, const Elem& y) { return ((x.v != x.v) && !(y.v != y.v)) || (x.v > y.v); }

// The equivalence class of a value under `gt` as an integer that DESCENDS with the value: all NaNs share the smallest key
// (they sort first), -0 and +0 share one, otherwise the usual order-preserving map of the float bits, inverted.
FP_HD unsigned class_key(const Elem& x) {
  if (x.v != x.v) return 0u;
  const float f = x.v == 0.f ? 0.f : x.v;
  unsigned b;
#if defined(__HIP_DEVICE_COMPILE__)
  b = __float_as_uint(f);
#else
  __builtin_memcpy(&b, &f, 4);
#endif
  b = (b & 0x80000000u) ? ~b : (b | 0x80000000u);  // ascending with the value; never 0xffffffff for a non-NaN
  return ~b;                                        // descending; > 0 for every non-NaN (+inf -> 0x007fffff)
}

FP_HD void swap_(Elem& a, Elem& b) {
  const Elem t = a;
  a = b;
  b = t;
}

FP_HD int lg2(int n) {  // floor(log2(n)), n >= 1
  int r = 0;
  while (n > 1) {
    n >>= 1;
    ++r;
  }
  return r;
}

// ---- heap primitives (max-heap w.r.t. `gt` as "less": the root is the WORST of the kept elements).  Written over an
// accessor (get / set by index) so the same moves run on an array in memory and on a heap whose element j lives in the
// registers of lane j of a wavefront (match.hip: the strict top-n keeps its n-element heap there).
struct PtrAcc {
  Elem* p;
  FP_HD Elem get(int i) const { return p[i]; }
  FP_HD void set(int i, const Elem& e) const { p[i] = e; }
};

template <class A>
FP_HD void push_heap_acc(A& a, int hole, int top, Elem value) {
  int parent = (hole - 1) / 2;
  while (hole > top && gt(a.get(parent), value)) {
    a.set(hole, a.get(parent));
    hole = parent;
    parent = (hole - 1) / 2;
  }
  a.set(hole, value);
}

template <class A>
FP_HD void adjust_heap_acc(A& a, int hole, int len, Elem value) {
  const int top = hole;
  int child = hole;
  while (child < (len - 1) / 2) {
    child = 2 * (child + 1);
    if (gt(a.get(child), a.get(child - 1))) --child;
    a.set(hole, a.get(child));
    hole = child;
  }
  if ((len & 1) == 0 && child == (len - 2) / 2) {
    child = 2 * (child + 1);
    a.set(hole, a.get(child - 1));
    hole = child - 1;
  }
  push_heap_acc(a, hole, top, value);
}

template <class A>
FP_HD void make_heap_acc(A& a, int len) {
  if (len < 2) return;
  int parent = (len - 2) / 2;
  while (true) {
    const Elem value = a.get(parent);
    adjust_heap_acc(a, parent, len, value);
    if (parent == 0) return;
    --parent;
  }
}

template <class A>
FP_HD void sort_heap_acc(A& a, int len) {
  while (len > 1) {
    --len;
    const Elem value = a.get(len);  // __pop_heap(first, last, last): the root moves to the end, the old end re-enters
    a.set(len, a.get(0));
    adjust_heap_acc(a, 0, len, value);
  }
}

FP_HD void push_heap_(Elem* a, int hole, int top, Elem value) {
  PtrAcc acc{a};
  push_heap_acc(acc, hole, top, value);
}
FP_HD void adjust_heap_(Elem* a, int hole, int len, Elem value) {
  PtrAcc acc{a};
  adjust_heap_acc(acc, hole, len, value);
}
FP_HD void make_heap_(Elem* a, int len) {
  PtrAcc acc{a};
  make_heap_acc(acc, len);
}

// pop the root of heap a[0, len) into *result (result may be outside the heap)
FP_HD void pop_heap_(Elem* a, int len, Elem* result) {
  const Elem value = *result;
  *result = a[0];
  adjust_heap_(a, 0, len, value);
}

FP_HD void heap_select_(Elem* a, int middle, int last) {
  make_heap_(a, middle);
  for (int i = middle; i < last; ++i)
    if (gt(a[i], a[0])) pop_heap_(a, middle, &a[i]);
}

FP_HD void sort_heap_(Elem* a, int len) {
  PtrAcc acc{a};
  sort_heap_acc(acc, len);
}

FP_HD void partial_sort_(Elem* a, int middle, int last) {
  heap_select_(a, middle, last);
  sort_heap_(a, middle);
}

// ---- quicksort pieces
FP_HD void move_median_to_first_(Elem* a, int result, int x, int y, int z) {
  if (gt(a[x], a[y])) {
    if (gt(a[y], a[z])) swap_(a[result], a[y]);
    else if (gt(a[x], a[z])) swap_(a[result], a[z]);
    else swap_(a[result], a[x]);
  } else if (gt(a[x], a[z])) swap_(a[result], a[x]);
  else if (gt(a[y], a[z])) swap_(a[result], a[z]);
  else swap_(a[result], a[y]);
}

FP_HD int unguarded_partition_(Elem* a, int first, int last, int pivot) {
  while (true) {
    while (gt(a[first], a[pivot])) ++first;
    --last;
    while (gt(a[pivot], a[last])) --last;
    if (!(first < last)) return first;
    swap_(a[first], a[last]);
    ++first;
  }
}

FP_HD int unguarded_partition_pivot_(Elem* a, int first, int last) {
  const int mid = first + (last - first) / 2;
  move_median_to_first_(a, first, first + 1, mid, last - 1);
  return unguarded_partition_(a, first + 1, last, first);
}

FP_HD void unguarded_linear_insert_(Elem* a, int last) {
  const Elem val = a[last];
  int next = last - 1;
  while (gt(val, a[next])) {
    a[last] = a[next];
    last = next;
    --next;
  }
  a[last] = val;
}

FP_HD void insertion_sort_(Elem* a, int first, int last) {
  if (first == last) return;
  for (int i = first + 1; i != last; ++i) {
    if (gt(a[i], a[first])) {
      const Elem val = a[i];
      for (int j = i; j > first; --j) a[j] = a[j - 1];
      a[first] = val;
    } else {
      unguarded_linear_insert_(a, i);
    }
  }
}

FP_HD void nth_element_(Elem* a, int first, int nth, int last) {
  if (first == last || nth == last) return;
  int depth = lg2(last - first) * 2;
  while (last - first > 3) {
    if (depth == 0) {
      heap_select_(a + first, nth + 1 - first, last - first);
      swap_(a[first], a[nth]);
      return;
    }
    --depth;
    const int cut = unguarded_partition_pivot_(a, first, last);
    if (cut <= nth) first = cut;
    else last = cut;
  }
  insertion_sort_(a, first, last);
}

// std::sort: introsort loop (recursion on the right part turned into an explicit stack) + final insertion sort
FP_HD void sort_(Elem* a, int first, int last) {
  if (first == last) return;
  struct Frame {
    int first, last, depth;
  };
  Frame stack[48];
  int sp = 0;
  stack[sp++] = Frame{first, last, lg2(last - first) * 2};
  while (sp > 0) {
    Frame f = stack[--sp];
    while (f.last - f.first > 16) {
      if (f.depth == 0) {
        partial_sort_(a + f.first, f.last - f.first, f.last - f.first);  // heapsort of the range
        break;
      }
      --f.depth;
      const int cut = unguarded_partition_pivot_(a, f.first, f.last);
      // libstdc++ recurses into [cut, last) first and then continues with [first, cut): ranges are disjoint, so
      // deferring the right part on a stack performs exactly the same element moves on each range.
      stack[sp++] = Frame{cut, f.last, f.depth};
      f.last = cut;
    }
  }
  if (last - first > 16) {
    insertion_sort_(a, first, first + 16);
    for (int i = first + 16; i != last; ++i) unguarded_linear_insert_(a, i);
  } else {
    insertion_sort_(a, first, last);
  }
}

// In place: afterwards a[0, k) holds torch.topk's output order.  a must be filled as a[j] = {x[j], j}.
FP_HD void topk_torch_largest(Elem* a, int n, int k) {
  if (k <= 0 || n <= 0) return;
  if ((long long)k * 64 <= (long long)n) {
    partial_sort_(a, k, n);
  } else {
    nth_element_(a, 0, k - 1, n);
    sort_(a, 0, k - 1);
  }
}

}  // namespace stl_order
