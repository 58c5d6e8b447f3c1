// Wave-parallel replay of the libstdc++ algorithms behind torch.topk (see stl_order.hpp for what is replayed and why).
//
// stl_order.hpp restates std::nth_element / std::sort element move for element move, but as sequential code: one lane
// walking an LDS array spends ~100 cycles per dependent step, and the best-buddy selection (top 300 of ~520 heavily tied
// cycle distances, utils/corresp_util.py:61) took 0.8 ms per batch that way.  The moves themselves are data-parallel:
//
//  * __unguarded_partition(first, last, pivot) swaps the m-th "left stopper" (element not before the pivot, scanning up)
//    with the m-th "right stopper" (element not after the pivot, scanning down) for m = 1, 2, ... while the former lies
//    left of the latter.  A swapped element is never looked at again, so both stopper sequences are those of the
//    ORIGINAL array: two ballots per 64 elements give them, a popcount prefix gives every stopper its rank, all swaps
//    happen at once, and the returned cut is the next left stopper if one lies before the last swapped right position,
//    else that position (where the scan would meet the element it just moved there).
//  * the final insertion sort of std::sort is a stable sort of whatever arrangement the quicksort phase left: every
//    element's final place = number of elements that must precede it, counted in parallel.
//
// One wave (64 lanes) executes these; control flow (ranges, recursion stack, depth limits, the heap fallbacks) is wave
// uniform and identical to stl_order.hpp, whose sequential pieces are reused for the cold paths.  LDS operations of one
// wave execute in order, so lanes exchange data through LDS with a compiler fence only.
#pragma once
#include "stl_order.hpp"

namespace stl_wave {
using stl_order::Elem;
using stl_order::gt;

__device__ __forceinline__ void wsync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// position of the n-th (0-based) set bit of m, counted from bit 0; n < popcount(m)
__device__ __forceinline__ int select_bit(unsigned long long m, int n) {
  int pos = 0;
#pragma unroll
  for (int w = 32; w >= 1; w >>= 1) {
    const int c = __popcll((m >> pos) & ((1ull << w) - 1ull));
    if (n >= c) {
      n -= c;
      pos += w;
    }
  }
  return pos;
}

// The same partition for a range whose scanned part a[lo + 1, hi) fits one chunk of 64 (most calls of an introsort over a
// few hundred elements): everything stays in registers -- the median-of-3 from four broadcast reads, the stopper ranks from
// two ballots, the partner of a swapped stopper by selecting the matching set bit of the other ballot, the exchange by one
// lane permute -- instead of rank tables in LDS and four dependent LDS round trips.
__device__ inline int partition_pivot_small(Elem* a, int lo, int hi, int lane) {
  const int mid = lo + (hi - lo) / 2;
  const Elem e0 = a[lo], ex = a[lo + 1], ey = a[mid], ez = a[hi - 1];
  int src;  // move_median_to_first_: whose element is swapped with a[lo]
  if (gt(ex, ey)) src = gt(ey, ez) ? mid : (gt(ex, ez) ? hi - 1 : lo + 1);
  else src = gt(ex, ez) ? lo + 1 : (gt(ey, ez) ? hi - 1 : mid);
  const Elem piv = src == lo + 1 ? ex : (src == mid ? ey : ez);
  const int i = lo + 1 + lane;
  const bool in = i < hi;
  Elem v = in ? a[i] : piv;
  if (i == src) v = e0;  // the arrangement after the median swap
  const bool fl = in && !gt(v, piv), fr = in && !gt(piv, v);
  const unsigned long long bl = __ballot(fl), br = __ballot(fr);
  const unsigned long long below = (1ull << lane) - 1ull;
  const int n_l = __popcll(bl), n_r = __popcll(br), np = n_l < n_r ? n_l : n_r;
  const int m_l = __popcll(bl & below);                       // rank among the left stoppers, from the left
  const int m_r = __popcll(br & ~below & ~(1ull << lane));    // rank among the right stoppers, from the right
  int partner = lane;
  bool sw_l = false;
  if (fl && m_l < np) {
    partner = select_bit(br, n_r - 1 - m_l);
    sw_l = lane < partner;
  }
  const int s = __popcll(__ballot(sw_l));  // swapped pairs: the condition is monotone in the rank
  if (!sw_l) partner = (fr && m_r < s) ? select_bit(bl, m_r) : lane;
  Elem nv;
  nv.v = __shfl(v.v, partner, 64);
  nv.idx = __shfl(v.idx, partner, 64);
  if (lane == 0) a[lo] = piv;
  if (in) a[i] = nv;
  wsync();
  const int next_l = s < n_l ? lo + 1 + select_bit(bl, s) : 0x7fffffff;
  const int last_r = s >= 1 ? lo + 1 + select_bit(br, n_r - s) : hi;
  return next_l < last_r ? next_l : last_r;
}

// __unguarded_partition_pivot(first = lo, last = hi) on a[lo, hi), hi - lo > 3.  lpos / rpos: scratch for hi - lo ranks.
__device__ inline int partition_pivot(Elem* a, int lo, int hi, unsigned short* lpos, unsigned short* rpos, int lane) {
  if (hi - lo - 1 <= 64) return partition_pivot_small(a, lo, hi, lane);
  const int mid = lo + (hi - lo) / 2;
  if (lane == 0) stl_order::move_median_to_first_(a, lo, lo + 1, mid, hi - 1);
  wsync();
  const Elem piv = a[lo];
  const unsigned long long below = (1ull << lane) - 1ull;
  int n_l = 0, n_r = 0;
  for (int c = lo + 1; c < hi; c += 64) {
    const int i = c + lane;
    const bool in = i < hi;
    const Elem v = in ? a[i] : piv;
    const bool fl = in && !gt(v, piv);   // stops the upward scan  `while (comp(*first, pivot)) ++first`
    const bool fr = in && !gt(piv, v);   // stops the downward scan `while (comp(pivot, *last)) --last`
    const unsigned long long bl = __ballot(fl), br = __ballot(fr);
    if (fl) lpos[n_l + __popcll(bl & below)] = (unsigned short)i;
    if (fr) rpos[n_r + __popcll(br & below)] = (unsigned short)i;
    n_l += __popcll(bl);
    n_r += __popcll(br);
  }
  wsync();
  // pair m (0-based): m-th left stopper from the left with the m-th right stopper from the right; swapped while left < right
  const int np = n_l < n_r ? n_l : n_r;
  int s = 0;
  for (int m0 = 0; m0 < np; m0 += 64) {
    const int m = m0 + lane;
    int l = 0, r = 0;
    bool sw = false;
    if (m < np) {
      l = lpos[m];
      r = rpos[n_r - 1 - m];
      sw = l < r;
    }
    const unsigned long long b = __ballot(sw);
    if (sw) {
      const Elem x = a[l], y = a[r];
      a[l] = y;
      a[r] = x;
    }
    s += __popcll(b);
    if (b != ~0ull) break;  // the condition is monotone in m
  }
  wsync();
  const int next_l = s < n_l ? (int)lpos[s] : 0x7fffffff;
  const int last_r = s >= 1 ? (int)rpos[n_r - s] : hi;
  return next_l < last_r ? next_l : last_r;
}

__device__ inline void nth_element(Elem* a, int first, int nth, int last, unsigned short* lpos, unsigned short* rpos, int lane) {
  if (first == last || nth == last) return;
  int depth = stl_order::lg2(last - first) * 2;
  while (last - first > 3) {
    if (depth == 0) {
      if (lane == 0) {
        stl_order::heap_select_(a + first, nth + 1 - first, last - first);
        stl_order::swap_(a[first], a[nth]);
      }
      wsync();
      return;
    }
    --depth;
    const int cut = partition_pivot(a, first, last, lpos, rpos, lane);
    if (cut <= nth) first = cut;
    else last = cut;
  }
  if (lane == 0) stl_order::insertion_sort_(a, first, last);
  wsync();
}

// std::sort(a + first, a + last).  tmp: scratch for last - first elements.
__device__ inline void sort(Elem* a, int first, int last, unsigned short* lpos, unsigned short* rpos, Elem* tmp, int lane) {
  if (last - first < 2) return;
  struct Frame {
    int first, last, depth;
  };
  Frame stack[48];
  int sp = 0;
  stack[sp++] = Frame{first, last, stl_order::lg2(last - first) * 2};
  while (sp > 0) {
    Frame f = stack[--sp];
    while (f.last - f.first > 16) {
      if (f.depth == 0) {
        if (lane == 0) stl_order::partial_sort_(a + f.first, f.last - f.first, f.last - f.first);  // heapsort of the range
        wsync();
        break;
      }
      --f.depth;
      const int cut = partition_pivot(a, f.first, f.last, lpos, rpos, lane);
      stack[sp++] = Frame{cut, f.last, f.depth};
      f.last = cut;
    }
  }
  // __final_insertion_sort == the stable sort of the current arrangement: place = #elements that precede.  "u precedes v"
  // (gt(u, v), or equivalent and earlier in the arrangement) is an order on (class of the value, position): the value's class as
  // a descending integer key (all NaNs one class, first; -0 = +0) with the position below it makes one u64 per element, and
  // place(i) = #{j : key_j < key_i} -- one LDS broadcast read and one 64-bit compare per pair instead of two float
  // comparisons with NaN logic (the selection kernel spent a quarter of its time here).
  const int n = last - first;
  unsigned long long* keys = reinterpret_cast<unsigned long long*>(tmp);  // n keys; the elements move through registers
  constexpr int MAXC = 8;  // chunks of 64 a lane keeps in registers: n <= 512 here (k - 1 <= 511), else the two-pass form below
  if (n <= 64 * MAXC) {
    Elem mine[MAXC];
    unsigned long long kmine[MAXC];
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      const int i = c * 64 + lane;
      if (i < n) {
        mine[c] = a[first + i];
        kmine[c] = ((unsigned long long)stl_order::class_key(mine[c]) << 32) | (unsigned)i;
        keys[i] = kmine[c];
      }
    }
    wsync();
    int place[MAXC];
#pragma unroll
    for (int c = 0; c < MAXC; ++c) place[c] = 0;
    for (int j = 0; j < n; ++j) {
      const unsigned long long kj = keys[j];  // same address on every lane: an LDS broadcast
#pragma unroll
      for (int c = 0; c < MAXC; ++c)
        if (c * 64 < n) place[c] += kj < kmine[c] ? 1 : 0;
    }
    wsync();  // all reads of keys done before the elements overwrite... (keys alias tmp, not a)
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      const int i = c * 64 + lane;
      if (i < n) a[first + place[c]] = mine[c];
    }
    wsync();
    return;
  }
  for (int i0 = 0; i0 < n; i0 += 64) {
    const int i = i0 + lane;
    if (i < n) {
      const Elem v = a[first + i];
      int place = 0;
      for (int j = 0; j < n; ++j) {
        const Elem u = a[first + j];  // same address on every lane: an LDS broadcast
        place += (gt(u, v) || (j < i && !gt(v, u))) ? 1 : 0;
      }
      tmp[place] = v;
    }
  }
  wsync();
  for (int i = lane; i < n; i += 64) a[first + i] = tmp[i];
  wsync();
}

// In place: afterwards a[0, k) holds torch.topk(x, k, largest=True, sorted=True)'s output order; a[j] = {x[j], j} on entry.
// n <= 65535 positions (unsigned short ranks); lpos / rpos hold n entries, tmp holds k.
__device__ inline void topk_torch_largest(Elem* a, int n, int k, unsigned short* lpos, unsigned short* rpos, Elem* tmp, int lane) {
  if (k <= 0 || n <= 0) return;
  if ((long long)k * 64 <= (long long)n) {
    if (lane == 0) stl_order::partial_sort_(a, k, n);
    wsync();
  } else {
    nth_element(a, 0, k - 1, n, lpos, rpos, lane);
    sort(a, 0, k - 1, lpos, rpos, tmp, lane);
  }
}

}  // namespace stl_wave
