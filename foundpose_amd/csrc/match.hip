// Descriptor-matching kernels that are not GEMM-shaped (HBM/latency-bound integer + fp32 work):
// row norms, canonical row top-k, tf-idf histogram, cyclic best-buddy selection, bilinear sampling.
//
// Reference behaviour restated (paths under /root/reference):
//   topk_rows ........ faiss heap result order (utils/knn_util.py:83) / torch.topk (utils/template_util.py:172)
//                      canonical order here: best value first, ties -> lowest index
//   tfidf_build ...... utils/template_util.py:31-71 (weights, L2-normalise per query, tf = w/Q,
//                      scatter_add_ in flattened order) + the query side of cosine_similarity (:167)
//   cyclic_select .... utils/corresp_util.py:49-70,135-155
//   sample_bilinear .. utils/feature_util.py:100-131 (grid_sample bilinear, zeros, align_corners=False)
#include <cstdlib>

#include "common.hpp"
#include "kernels.hpp"
#include "stl_order.hpp"

namespace {

// ------------------------------------------------------------------ |x|^2 per row, k-ascending fmaf chain
__global__ void sqnorm_rows_kernel(const float* __restrict__ x, long long n, int d, int ld, float* __restrict__ out) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4* r = reinterpret_cast<const float4*>(x + i * ld);
  float acc = 0.f;
  for (int k = 0; k < d / 4; ++k) {
    float4 v = r[k];
    acc = fmaf(v.x, v.x, acc);
    acc = fmaf(v.y, v.y, acc);
    acc = fmaf(v.z, v.z, acc);
    acc = fmaf(v.w, v.w, acc);
  }
  out[i] = acc;
}

// x / max(sqrt(|x|^2), eps) per row (bank-side cosine normalisation; |x|^2 as above)
__global__ void normalize_rows_kernel(const float* __restrict__ x, long long n, int d, float eps, float* __restrict__ out) {
  // one wave per row; lane 0 computes the chain so the order matches the oracle
  long long row = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= n) return;
  const int lane = threadIdx.x & 63;
  const float* r = x + row * d;
  float acc = 0.f;
  if (lane == 0)
    for (int k = 0; k < d; ++k) acc = fmaf(r[k], r[k], acc);
  acc = __shfl(acc, 0, 64);
  const float nrm = fmaxf(sqrtf(acc), eps);
  for (int k = lane; k < d; k += 64) out[row * d + k] = r[k] / nrm;
}

// ------------------------------------------------------------------ canonical top-k along rows
FP_DEVICE unsigned order_key(float v, bool largest) {
  unsigned b = __float_as_uint(v);
  b = (b & 0x80000000u) ? ~b : (b | 0x80000000u);  // ascending float order as unsigned
  return largest ? ~b : b;
}

// One wave per row. k selection passes; each pass takes the smallest (key, index) above the previous one.
__global__ void topk_rows_kernel(const float* __restrict__ vals, int rows, int n, int ld, const int* __restrict__ row_len,
                                 int k, int largest, float* __restrict__ out_val, int* __restrict__ out_idx) {
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const int len = row_len ? row_len[row] : n;
  const float* r = vals + (size_t)row * ld;
  unsigned long long prev = 0;
  bool have_prev = false;
  for (int s = 0; s < k; ++s) {
    unsigned long long best = ~0ull;
    for (int j = lane; j < len; j += 64) {
      unsigned long long key = ((unsigned long long)order_key(r[j], largest) << 32) | (unsigned)j;
      if ((!have_prev || key > prev) && key < best) best = key;
    }
    best = wave_min_u64(best);
    if (lane == 0) {
      if (best != ~0ull) {
        int j = (int)(best & 0xffffffffu);
        out_idx[(size_t)row * k + s] = j;
        out_val[(size_t)row * k + s] = r[j];
      } else {
        out_idx[(size_t)row * k + s] = -1;
        out_val[(size_t)row * k + s] = largest ? -INFINITY : INFINITY;
      }
    }
    prev = best;
    have_prev = true;
  }
}

// ------------------------------------------------------------------ template retrieval: bank-streaming cosine scores
// sims[det][t] = <bank_n[t,:], q_n[det,:]> for every template t of the detection's object.
// HBM-bound by design: each wave owns 16 template rows and streams them ONCE, straight from HBM into registers
// (16 B per lane, no LDS: the rows are not shared between waves), against all <= 64 detections of the object
// (query rows come from L2).  v_mfma_f32_16x16x4_f32: A = 16 templates x 4 k, B = 4 k x 16 detections.
// A lane's float4 covers k = 16j + 4g .. +3 (g = lane>>4), so MFMA step u consumes k = 16j + 4g' + u, g' = 0..3:
// the per-(template, detection) fp32 fma chain visits each 16-block of k in the order
// [0,4,8,12, 1,5,9,13, 2,6,10,14, 3,7,11,15] -- the canonical order of this stage (oracle: orc_dot_rows_perm16).
// A-operand (bank) loads of one chunk of U 16-blocks: 16 B per lane, non-temporal (streamed once).
template <int U>
FP_DEVICE void cos_load_a(f32x4 (&av)[U], const float* ap, int chunk) {
#pragma unroll
  for (int u = 0; u < U; ++u) av[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(ap + 16 * (chunk * U + u)));
}
// MFMAs of one chunk; the query fragments come from the LDS slice image (ds_read_b128, conflict-free: row pitch 1040 B).
// Accumulators alternate between consecutive MFMAs (dependent latency of 16x16x4 f32 > its issue interval).
template <int NQ, int U>
FP_DEVICE void cos_mma(f32x4 (&acc)[NQ], const f32x4 (&av)[U], const char* qs, int chunk, int pitch) {
#pragma unroll
  for (int u = 0; u < U; ++u) {
    float4 bv[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) bv[q] = *reinterpret_cast<const float4*>(qs + q * 16 * pitch + (chunk * U + u) * 64);
#pragma unroll
    for (int q = 0; q < NQ; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][0], bv[q].x, acc[q], 0, 0, 0);
#pragma unroll
    for (int q = 0; q < NQ; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][1], bv[q].y, acc[q], 0, 0, 0);
#pragma unroll
    for (int q = 0; q < NQ; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][2], bv[q].z, acc[q], 0, 0, 0);
#pragma unroll
    for (int q = 0; q < NQ; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][3], bv[q].w, acc[q], 0, 0, 0);
  }
}

template <int NQ>
__global__ __launch_bounds__(256) void cosine_sims_kernel(CosineArgs a) {
  extern __shared__ __attribute__((aligned(16))) char qlds[];  // [NQ*16 detections][wslice floats + 16 B pad]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int obj = blockIdx.y;
  const int tb = a.obj_tpl_off[obj], T = a.obj_tpl_off[obj + 1] - tb;
  const int d0 = a.det_seg_off[obj], nd = a.det_seg_off[obj + 1] - d0;
  if (blockIdx.x * 64 >= T || nd <= 0) return;  // block-uniform
  // K is cut into a.k_slices contiguous slices (blockIdx.z): more waves than SIMDs, so the fp32 matrix pipe of
  // every SIMD works on the stream; the slice chains are summed in slice order by the top-n kernel.
  const int kslice = blockIdx.z, wslice = a.W / a.k_slices;
  const int pitch = wslice * 4 + 16;
  // ---- the object's query descriptors (this k-slice) go to LDS once per block, as whole rows (1 KiB per wave
  // instruction at wslice = 256): per-wave register loads of them cost 2/3 of the kernel's load instructions and the
  // address path, not HBM, became the limit (profiles/r1_pmc_counters.txt)
  for (int r = wave; r < NQ * 16; r += 4) {
    const float* src = a.desc_n + (size_t)(d0 + min(r, nd - 1)) * a.W + kslice * wslice;
    for (int c = lane * 4; c < wslice; c += 256)
      *reinterpret_cast<float4*>(qlds + r * pitch + c * 4) = *reinterpret_cast<const float4*>(src + c);
  }
  const int t0 = (blockIdx.x * 4 + wave) * 16;
  const int i = lane & 15, g = lane >> 4;
  const int trow = min(t0 + i, T - 1);
  const float* ap = a.bank_n + (size_t)(tb + trow) * a.W + kslice * wslice + 4 * g;
  const char* qs = qlds + i * pitch + g * 16;
  f32x4 acc[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nb = wslice / 16;
  // Software pipeline: chunks of U 16-blocks; chunk c+1's bank loads are in flight while chunk c's MFMAs run
  // (a wave keeps 2 x U KiB of the bank stream outstanding -- what it takes to pull HBM bandwidth without LDS).
  constexpr int U = 8;
  f32x4 a0[U], a1[U];
  const int nch = nb / U;
  const bool active = t0 < T;
  if (active && nch > 0) cos_load_a<U>(a0, ap, 0);
  __syncthreads();  // query slice staged
  if (!active) return;
  for (int c = 0; c < nch; c += 2) {
    if (c + 1 < nch) cos_load_a<U>(a1, ap, c + 1);
    cos_mma<NQ, U>(acc, a0, qs, c, pitch);
    if (c + 2 < nch) cos_load_a<U>(a0, ap, c + 2);
    if (c + 1 < nch) cos_mma<NQ, U>(acc, a1, qs, c + 1, pitch);
  }
  for (int j = nch * U; j < nb; ++j) {  // remainder blocks (slice not a multiple of 128 floats)
    const f32x4 av = *reinterpret_cast<const f32x4*>(ap + 16 * j);
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const float4 bv = *reinterpret_cast<const float4*>(qs + q * 16 * pitch + j * 64);
      acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0], bv.x, acc[q], 0, 0, 0);
      acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1], bv.y, acc[q], 0, 0, 0);
      acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[2], bv.z, acc[q], 0, 0, 0);
      acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[3], bv.w, acc[q], 0, 0, 0);
    }
  }
  // D[i = template 4g + r][j = detection lane&15]
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int det = q * 16 + i;
    if (det >= nd) continue;
    float* o = a.sims + (size_t)kslice * a.slice_stride + (size_t)(d0 + det) * a.ld_sims + t0 + 4 * g;
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (t0 + 4 * g + r < T) o[r] = acc[q][r];
  }
}

// Same arithmetic, bank rows staged through LDS by DMA.  The register-direct kernel above reads 16 B per lane from 16
// different template rows per load instruction (the MFMA operand layout): 64 B-per-row accesses cost the address path
// four times what whole rows cost, and that -- not HBM -- bounded it at ~2.5 TB/s.  Here one persistent 8-wave
// workgroup per CU serves one (object, k-slice); each wave streams its 16-template blocks as 4-KiB chunks (16 rows x
// 256 B, fetched as whole 256-B row segments by four global_load_lds) through a private three-slot LDS ring: two
// chunks (8 KiB per wave, 64 KiB per CU) are always in flight under the MFMAs of the third, waits are counted vmcnt,
// and no barrier is needed because a wave only reads what it fetched itself.  The 16-B pieces of a row are XOR-placed
// by the row index (on the DMA source address) so the per-lane fragment reads hit 16 different bank groups.
// Fragment contents and MFMA order are those of cosine_sims_kernel: bit-identical scores.
typedef __attribute__((address_space(3))) void cos_lds_void;
typedef __attribute__((address_space(1))) const void cos_gbl_cvoid;

template <int NQ>
__global__ __launch_bounds__(1024) void cosine_stream_kernel(CosineArgs a) {
  extern __shared__ __attribute__((aligned(16))) char qlds[];  // [NQ*16 detections][wslice floats + 16 B] | [waves][3][4 KiB]
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int obj = blockIdx.y;
  const int tb = a.obj_tpl_off[obj], T = a.obj_tpl_off[obj + 1] - tb;
  const int d0 = a.det_seg_off[obj], nd = a.det_seg_off[obj + 1] - d0;
  if (T <= 0 || nd <= 0) return;  // block-uniform
  const int kslice = blockIdx.z, wslice = a.W / a.k_slices;
  const int pitch = wslice * 4 + 16;
  const int nwave = blockDim.x >> 6;
  for (int r = wave; r < NQ * 16; r += nwave) {
    const float* src = a.desc_n + (size_t)(d0 + min(r, nd - 1)) * a.W + kslice * wslice;
    for (int c = lane * 4; c < wslice; c += 256)
      *reinterpret_cast<float4*>(qlds + r * pitch + c * 4) = *reinterpret_cast<const float4*>(src + c);
  }
  char* ring = qlds + NQ * 16 * pitch + wave * (3 * 4096);
  const int nblk = (T + 15) >> 4, nch = wslice >> 6;         // 16-template blocks of the object, 64-word chunks per block
  const int first = blockIdx.x * nwave + wave, stride = gridDim.x * nwave;
  const int ntask = first < nblk ? (nblk - first + stride - 1) / stride : 0;
  const int total = ntask * nch;
  __syncthreads();  // query slice staged (the only barrier)
  if (total == 0) return;

  const int i = lane & 15, g = lane >> 4;
  const float* slice_base = a.bank_n + (size_t)tb * a.W + kslice * wslice;
  // issue side: (task, chunk) cursor of the next chunk to fetch
  int it_blk = first, it_ch = 0, it_slot = 0;
  auto issue = [&]() {
    const int t0 = it_blk * 16;
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      const int row = 4 * q4 + g;                     // lane -> (row of the block, physical 16-B slot i)
      const int piece = i ^ row;                      // logical piece that must land in slot i of this row
      const float* src = slice_base + (size_t)min(t0 + row, T - 1) * a.W + it_ch * 64 + piece * 4;
      __builtin_amdgcn_global_load_lds((cos_gbl_cvoid*)src, (cos_lds_void*)(ring + it_slot * 4096 + q4 * 1024), 16, 0, 2 /* nt */);
    }
    if (++it_ch == nch) { it_ch = 0; it_blk += stride; }
    it_slot = it_slot == 2 ? 0 : it_slot + 1;
  };
  issue();
  if (total > 1) issue();
  if (total > 2) issue();

  f32x4 acc[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
  const char* qs = qlds + i * pitch + g * 16;
  int blk = first, ch = 0, slot = 0;
  for (int cc = 0; cc < total; ++cc) {
    // loads return in order: chunk cc has landed once at most the later chunks' DMAs are outstanding
    if (cc + 2 < total) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (cc + 1 < total) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // the chunk moves to registers at once, so its slot can be refilled now: all three ring slots (12 KiB per wave,
    // 96 KiB per CU) are in flight while the MFMAs below run
    const char* cs = ring + slot * 4096 + i * 256;
    f32x4 av[4];
    float4 bv[4][NQ];
#pragma unroll
    for (int j = 0; j < 4; ++j) av[j] = *reinterpret_cast<const f32x4*>(cs + (((4 * j + g) ^ i) << 4));
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int q = 0; q < NQ; ++q) bv[j][q] = *reinterpret_cast<const float4*>(qs + q * 16 * pitch + (ch * 4 + j) * 64);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // fragments are in registers before the slot is handed back
    if (cc + 3 < total) issue();
#pragma unroll
    for (int j = 0; j < 4; ++j) {  // 32 (NQ = 2) back-to-back MFMAs, two alternating accumulator chains
#pragma unroll
      for (int q = 0; q < NQ; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j][0], bv[j][q].x, acc[q], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < NQ; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j][1], bv[j][q].y, acc[q], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < NQ; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j][2], bv[j][q].z, acc[q], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < NQ; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j][3], bv[j][q].w, acc[q], 0, 0, 0);
    }
    slot = slot == 2 ? 0 : slot + 1;
    if (++ch == nch) {
      // D[i = template 4g + r][j = detection lane&15]
      const int t0 = blk * 16;
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int det = q * 16 + i;
        if (det < nd) {
          // a lane holds 4 consecutive templates of one detection: one 16-B store when the row pitch allows it
          float* o = a.sims + (size_t)kslice * a.slice_stride + (size_t)(d0 + det) * a.ld_sims + t0 + 4 * g;
          if (t0 + 16 <= T && (a.ld_sims & 3) == 0) {
            *reinterpret_cast<f32x4*>(o) = acc[q];
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (t0 + 4 * g + r < T) o[r] = acc[q][r];
          }
        }
        acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      ch = 0;
      blk += stride;
    }
  }
}

// Canonical top-n of each row (largest first, ties -> lowest index), one 256-thread block per row:
// per-thread top-n over a strided slice (registers), then n rounds of block-wide arg-best over the candidates.
template <int NMAX>
__global__ __launch_bounds__(256) void topn_rows_block_kernel(float* __restrict__ vals, int ld, const int* __restrict__ row_len,
                                                              int n_default, int n_top, float* __restrict__ out_val, int* __restrict__ out_idx,
                                                              int k_slices, long long slice_stride, unsigned long long* __restrict__ cand_out) {
  __shared__ unsigned long long cand[256 * NMAX];
  __shared__ unsigned long long wbest[4];
  const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int len = row_len ? row_len[row] : n_default;
  float* r = vals + (size_t)row * ld;
  // gridDim.y > 1: this block scans one contiguous split of the row and emits its n_top best keys (phase A);
  // a merge launch (gridDim.y == 1 over the candidate keys) finishes the row.
  const int nsplit = gridDim.y, split = blockIdx.y;
  const int per = (len + nsplit - 1) / nsplit;
  const int j_begin = split * per, j_end = min(len, j_begin + per);
  unsigned long long best[NMAX];
#pragma unroll
  for (int s = 0; s < NMAX; ++s) best[s] = ~0ull;
  for (int j = j_begin + tid; j < j_end; j += 256) {
    float v = r[j];
    for (int sl = 1; sl < k_slices; ++sl) v += r[(size_t)sl * slice_stride + j];  // slice chains added in slice order
    unsigned long long key = ((unsigned long long)order_key(v, true) << 32) | (unsigned)j;
#pragma unroll
    for (int s = 0; s < NMAX; ++s) {  // sorted insertion (ascending keys = best first)
      const unsigned long long lo = key < best[s] ? key : best[s];
      key = key < best[s] ? best[s] : key;
      best[s] = lo;
    }
  }
#pragma unroll
  for (int s = 0; s < NMAX; ++s) cand[tid * NMAX + s] = best[s];
  __syncthreads();
  unsigned long long prev = 0;
  for (int s = 0; s < n_top; ++s) {
    unsigned long long b = ~0ull;
    for (int c = tid; c < 256 * NMAX; c += 256) {
      const unsigned long long k = cand[c];
      if ((s == 0 || k > prev) && k < b) b = k;
    }
    b = wave_min_u64(b);
    if (lane == 0) wbest[wave] = b;
    __syncthreads();
    b = wbest[0];
#pragma unroll
    for (int w = 1; w < 4; ++w) b = wbest[w] < b ? wbest[w] : b;
    if (tid == 0 && cand_out) {  // phase A of the split top-n: key + finished score of this split's s-th best
      const size_t slot = ((size_t)row * nsplit + split) * n_top + s;
      cand_out[slot] = b;
      float v = -INFINITY;
      if (b != ~0ull) {
        const int j = (int)(b & 0xffffffffu);
        v = r[j];
        for (int sl = 1; sl < k_slices; ++sl) v += r[(size_t)sl * slice_stride + j];
      }
      reinterpret_cast<float*>(cand_out + (size_t)gridDim.x * nsplit * n_top)[slot] = v;
    } else if (tid == 0) {
      if (b != ~0ull) {
        const int j = (int)(b & 0xffffffffu);
        float v = r[j];
        for (int sl = 1; sl < k_slices; ++sl) v += r[(size_t)sl * slice_stride + j];
        out_idx[(size_t)row * n_top + s] = j;
        out_val[(size_t)row * n_top + s] = v;
      } else {
        out_idx[(size_t)row * n_top + s] = -1;
        out_val[(size_t)row * n_top + s] = -INFINITY;
      }
    }
    prev = b;
    __syncthreads();
  }
}

// Phase A of the split top-n, one WAVE per (row, split): per-lane sorted top-n over a contiguous split (slice chains
// summed in slice order), then n rounds of wave-wide arg-best over the lanes' heads -- registers and lane shuffles only
// (the block-wide version above spent its time in 5 x (LDS scan + barrier) rounds).
template <int NMAX>
__global__ __launch_bounds__(256) void topn_rows_wave_kernel(const float* __restrict__ vals, int ld, const int* __restrict__ row_len,
                                                             int n_default, int n_top, int k_slices, long long slice_stride, int nsplit,
                                                             int rows, unsigned long long* __restrict__ cand_out) {
  const int row = blockIdx.x, lane = threadIdx.x & 63, split = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (split >= nsplit) return;
  const int len = row_len ? row_len[row] : n_default;
  const float* r = vals + (size_t)row * ld;
  const int per = (len + nsplit - 1) / nsplit;
  const int j_begin = split * per, j_end = min(len, j_begin + per);
  unsigned long long best[NMAX];
#pragma unroll
  for (int s = 0; s < NMAX; ++s) best[s] = ~0ull;
  // the partial scores were written by other XCDs, so every load here is an Infinity-Cache round trip (~1 us): batches
  // of 4 elements per lane put 4 x k_slices independent loads in flight before the first add
  for (int j0 = j_begin + lane; j0 < j_end; j0 += 256) {
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = j0 + 64 * e < j_end ? r[j0 + 64 * e] : 0.f;
    for (int sl = 1; sl < k_slices; ++sl) {  // slice chains added in slice order
      float w[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) w[e] = j0 + 64 * e < j_end ? r[(size_t)sl * slice_stride + j0 + 64 * e] : 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] += w[e];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      unsigned long long key = j0 + 64 * e < j_end ? ((unsigned long long)order_key(v[e], true) << 32) | (unsigned)(j0 + 64 * e) : ~0ull;
#pragma unroll
      for (int s = 0; s < NMAX; ++s) {  // sorted insertion (ascending keys = best first)
        const unsigned long long lo = key < best[s] ? key : best[s];
        key = key < best[s] ? best[s] : key;
        best[s] = lo;
      }
    }
  }
  float* cval = reinterpret_cast<float*>(cand_out + (size_t)rows * nsplit * n_top);
  for (int s = 0; s < n_top; ++s) {
    const unsigned long long b = wave_min_u64(best[0]);
    const size_t slot = ((size_t)row * nsplit + split) * n_top + s;
    if (b == ~0ull) {
      if (lane == 0) { cand_out[slot] = b; cval[slot] = -INFINITY; }
    } else if (best[0] == b) {  // exactly one lane owns the winner (keys carry the unique column index)
      const unsigned kb = ~(unsigned)(b >> 32);  // order_key inverted: the score's own bits, no reload
      cand_out[slot] = b;
      cval[slot] = __uint_as_float((kb & 0x80000000u) ? (kb ^ 0x80000000u) : ~kb);
#pragma unroll
      for (int t = 0; t + 1 < NMAX; ++t) best[t] = best[t + 1];
      best[NMAX - 1] = ~0ull;
    }
  }
}

// Phase B of the split top-n: one wave per row merges the nsplit * n_top (<= 128) candidates held in registers.
__global__ void topn_merge_kernel(const unsigned long long* __restrict__ cand, int ncand, int rows, int n_top,
                                  float* __restrict__ out_val, int* __restrict__ out_idx) {
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const unsigned long long* c = cand + (size_t)row * ncand;
  const float* cv = reinterpret_cast<const float*>(cand + (size_t)rows * ncand) + (size_t)row * ncand;
  unsigned long long k0 = lane < ncand ? c[lane] : ~0ull, k1 = lane + 64 < ncand ? c[lane + 64] : ~0ull;
  const float v0 = lane < ncand ? cv[lane] : 0.f, v1 = lane + 64 < ncand ? cv[lane + 64] : 0.f;
  for (int s = 0; s < n_top; ++s) {
    const unsigned long long mine = k0 < k1 ? k0 : k1;
    const unsigned long long b = wave_min_u64(mine);
    if (b != ~0ull && mine == b) {  // exactly one lane owns the winner (keys carry the unique column index)
      out_idx[(size_t)row * n_top + s] = (int)(b & 0xffffffffu);
      out_val[(size_t)row * n_top + s] = (k0 == b) ? v0 : v1;
      if (k0 == b) k0 = ~0ull; else k1 = ~0ull;
    }
    if (b == ~0ull && lane == 0) {
      out_idx[(size_t)row * n_top + s] = -1;
      out_val[(size_t)row * n_top + s] = -INFINITY;
    }
  }
}

// Strict-order variant: one block per row, the row staged in LDS as (value, index) pairs, one lane replays
// torch.topk's CPU algorithm (stl_order.hpp).  Rows of up to 20000 elements (160 KiB of LDS).
__global__ __launch_bounds__(256) void topn_rows_stl_kernel(float* __restrict__ vals, int ld, const int* __restrict__ row_len,
                                                            int n_default, int n_top, float* __restrict__ out_val, int* __restrict__ out_idx,
                                                            int k_slices, long long slice_stride) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  stl_order::Elem* el = reinterpret_cast<stl_order::Elem*>(smem_raw);
  const int row = blockIdx.x, tid = threadIdx.x;
  const int len = row_len ? row_len[row] : n_default;
  float* r = vals + (size_t)row * ld;
  for (int j = tid; j < len; j += 256) {
    float v = r[j];
    for (int sl = 1; sl < k_slices; ++sl) v += r[(size_t)sl * slice_stride + j];
    el[j] = stl_order::Elem{v, j};
  }
  __syncthreads();
  const int k = min(n_top, len);
  if (tid == 0) stl_order::topk_torch_largest(el, len, k);
  __syncthreads();
  if (tid < n_top) {
    out_idx[(size_t)row * n_top + tid] = tid < k ? el[tid].idx : -1;
    out_val[(size_t)row * n_top + tid] = tid < k ? el[tid].v : -INFINITY;
  }
}

// ------------------------------------------------------------------ tf-idf descriptor per detection
// One block (256 threads) per segment (a detection's query patches, or a template's patches on the
// bank-builder side). Thread t owns the bins with (id & 255) == t and walks the (q, j) entries in
// flattened order, so every bin sees its addends in exactly the order scatter_add_ applies them.
__global__ __launch_bounds__(256) void tfidf_build_kernel(
    const int* __restrict__ word_ids, const float* __restrict__ word_d2, int knn_k, const int* __restrict__ seg_off,
    const float* __restrict__ idf, int num_words, int soft, float two_sigma_sq, int sqrt_dists,
    float* __restrict__ desc, float* __restrict__ desc_n, float eps) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* bins = reinterpret_cast<float*>(smem_raw);            // [num_words]
  int* ids = reinterpret_cast<int*>(bins + num_words);         // [chunk]
  float* vals = reinterpret_cast<float*>(ids + 4096);          // [chunk]
  __shared__ float s_nrm;

  const int seg = blockIdx.x, tid = threadIdx.x;
  const int q0 = seg_off[seg], Q = seg_off[seg + 1] - q0;
  for (int w = tid; w < num_words; w += 256) bins[w] = 0.f;
  const float fQ = (float)Q;
  const int total = Q * knn_k;
  for (int base = 0; base < total; base += 4096) {
    __syncthreads();
    const int cnt = min(4096, total - base);
    // stage (id, tf*idf) for entries [base, base+cnt); entry e -> query e / k, neighbour e % k
    for (int e = tid; e < cnt; e += 256) {
      const int g = base + e;
      const int q = g / knn_k, j = g - q * knn_k;
      const int* idr = word_ids + (size_t)(q0 + q) * knn_k;
      const float* dr = word_d2 + (size_t)(q0 + q) * knn_k;
      float nrm2 = 0.f, wj = 1.f;
      for (int t = 0; t < knn_k; ++t) {
        float w = 1.f;
        if (soft) {
          float x = sqrt_dists ? sqrtf(dr[t]) : dr[t];
          w = expf(-(x * x) / two_sigma_sq);
        }
        nrm2 = nrm2 + w * w;  // sequential, like a 3-element fp32 sum
        if (t == j) wj = w;
      }
      const float wn = wj / fmaxf(sqrtf(nrm2), 1e-12f);
      const float tf = wn / fQ;
      const int id = idr[j];
      ids[e] = id;
      vals[e] = tf * idf[id];
    }
    __syncthreads();
    for (int e = 0; e < cnt; ++e) {
      const int id = ids[e];
      if ((id & 255) == tid) bins[id] += vals[e];
    }
  }
  __syncthreads();
  if (tid == 0) {  // |desc|^2 as ONE k-ascending fmaf chain (the canonical order); bins fetched 16 at a time so only the
    float acc = 0.f;  // fma latency, not an LDS round trip per element, is serial
    int w = 0;
    for (; w + 16 <= num_words; w += 16) {
      float4 b[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) b[u] = *reinterpret_cast<const float4*>(bins + w + 4 * u);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        acc = fmaf(b[u].x, b[u].x, acc); acc = fmaf(b[u].y, b[u].y, acc);
        acc = fmaf(b[u].z, b[u].z, acc); acc = fmaf(b[u].w, b[u].w, acc);
      }
    }
    for (; w < num_words; ++w) acc = fmaf(bins[w], bins[w], acc);
    s_nrm = fmaxf(sqrtf(acc), eps);
  }
  __syncthreads();
  const float nrm = s_nrm;
  for (int w = tid; w < num_words; w += 256) {
    const float v = bins[w];
    desc[(size_t)seg * num_words + w] = v;
    if (desc_n) desc_n[(size_t)seg * num_words + w] = v / nrm;
  }
}

// ------------------------------------------------------------------ cyclic best buddies: select + gather
// |u1 - u2|_2 exactly as separate fp32 mul/add + correctly rounded sqrt (no fma contraction), so the
// heavily tied cycle distances compare bit-for-bit with the CPU.
FP_DEVICE float point_dist(float x1, float y1, float x2, float y2) {
#pragma clang fp contract(off)
  const float dx = x1 - x2, dy = y1 - y2;
  const float xx = dx * dx, yy = dy * dy;
  return sqrtf(xx + yy);
}

__global__ __launch_bounds__(256) void cyclic_select_kernel(CyclicArgs a) {
  __shared__ unsigned long long keys[2048];
  __shared__ int q2o_s[2048];
  const int pair = blockIdx.x, tid = threadIdx.x;
  const int det = pair / a.n_slots;
  const int q0 = a.q_off[det], Q = a.q_off[det + 1] - q0;
  const int tpl = a.tpl_ids[pair];
  const int kk = min(a.top_k, Q);
  if (tid == 0) a.out_count[pair] = (tpl >= 0) ? kk : 0;
  if (tpl < 0 || Q == 0) return;
  const int f0 = a.tpl_off[tpl];
  const unsigned long long* rb = a.row_best + (size_t)pair * a.row_stride;
  const unsigned long long* cb = a.col_best + (size_t)pair * a.col_stride;
  const float* pts = a.points + (size_t)q0 * 2;

  for (int i = tid; i < 2048; i += 256) {
    unsigned long long key = ~0ull;
    if (i < Q) {
      const int o = (int)(rb[i] & 0xffffffffu);        // query -> nearest template patch
      const int c = (int)(cb[o] & 0xffffffffu);        // that patch -> nearest query patch
      q2o_s[i] = o;
      key = pack_dist_idx(point_dist(pts[2 * i], pts[2 * i + 1], pts[2 * c], pts[2 * c + 1]), (unsigned)i);
    }
    keys[i] = key;
  }
  __syncthreads();
  if (a.tie_mode == 1) {
    // strict mode: the reference's torch.topk(-cycle_dists, k) order, ties included -- one lane replays
    // libstdc++'s nth_element + sort (or partial_sort) on (value, index) pairs held in LDS
    stl_order::Elem* el = reinterpret_cast<stl_order::Elem*>(keys);
    for (int i = tid; i < Q; i += 256) {
      const unsigned long long key = keys[i];  // each thread converts only the slots it reads itself
      el[i] = stl_order::Elem{-__uint_as_float((unsigned)(key >> 32)), i};
    }
    __syncthreads();
    if (tid == 0) stl_order::topk_torch_largest(el, Q, kk);
    __syncthreads();
    for (int i = tid; i < kk; i += 256) {  // back to (distance bits, query id) keys in output order
      const stl_order::Elem e = el[i];
      keys[i] = pack_dist_idx(-e.v, (unsigned)e.idx);
    }
    __syncthreads();
  } else {
  // bitonic sort of 2048 keys, ascending: (cycle distance, query index)
  for (int size = 2; size <= 2048; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = tid; t < 1024; t += 256) {
        const int lo = 2 * t - (t & (stride - 1));
        const int hi = lo + stride;
        const bool up = ((lo & size) == 0);
        unsigned long long x = keys[lo], y = keys[hi];
        if ((x > y) == up) { keys[lo] = y; keys[hi] = x; }
      }
      __syncthreads();
    }
  }
  }
  const float dmax = __uint_as_float((unsigned)(keys[kk - 1] >> 32));
  const size_t ob = (size_t)pair * a.k_max;
  for (int r = tid; r < kk; r += 256) {
    const unsigned long long key = keys[r];
    const int qi = (int)(key & 0xffffffffu);
    const float d = __uint_as_float((unsigned)(key >> 32));
    const int feat = f0 + q2o_s[qi];
    a.out_q_ids[ob + r] = qi;
    a.out_feat_ids[ob + r] = feat - a.feat_base[det];
    a.out_dists[ob + r] = d;
    a.out_conf[ob + r] = 1.0f - d / dmax;
    a.out_coord_2d[(ob + r) * 2 + 0] = pts[2 * qi];
    a.out_coord_2d[(ob + r) * 2 + 1] = pts[2 * qi + 1];
    const float* v = a.vertices + (size_t)feat * 3;
    a.out_coord_3d[(ob + r) * 3 + 0] = v[0];
    a.out_coord_3d[(ob + r) * 3 + 1] = v[1];
    a.out_coord_3d[(ob + r) * 3 + 2] = v[2];
  }
}

// ------------------------------------------------------------------ bilinear sampling of a feature map
// One wave per point. fmap addressed through element strides so both the token-major [Np, D] image the
// extractor produces (a CHW *view*, like the reference's) and a contiguous CHW tensor work.
__global__ void sample_bilinear_kernel(SampleArgs a) {
  const int p = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (p >= a.num_points) return;
  const int lane = threadIdx.x & 63;
  const int img = a.point_img ? a.point_img[p] : 0;
  const float px = a.points[2 * p], py = a.points[2 * p + 1];
  // uv = (2/size) * p - 1   (fp32, no fma: feature_util.py:119)
  // Rounding sequence of torch's CPU grid_sampler (vectorised kernel, fma-contracted), verified
  // bit-for-bit against torch on the fixtures:  uv = (2/size)*p - 1 (separate mul, sub);
  // ix = fma(u + 1, W/2, -0.5);  w = ix - floor(ix), e = 1 - w;  out = fma chain over nw, ne, sw, se.
  float ix, iy;
  {
#pragma clang fp contract(off)
    const float sx = 2.0f / (float)a.img_w, sy = 2.0f / (float)a.img_h;
    const float ux = sx * px, uy = sy * py;
    const float u1 = (ux - 1.0f) + 1.0f, v1 = (uy - 1.0f) + 1.0f;
    ix = fmaf(u1, (float)a.W / 2.f, -0.5f);
    iy = fmaf(v1, (float)a.H / 2.f, -0.5f);
  }
  const float fx = floorf(ix), fy = floorf(iy);
  const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
  float w_nw, w_ne, w_sw, w_se;
  {
#pragma clang fp contract(off)
    const float w = ix - fx, e = 1.f - w, n = iy - fy, s_ = 1.f - n;
    w_nw = s_ * e; w_ne = s_ * w; w_sw = n * e; w_se = n * w;
  }
  const bool vx0 = x0 >= 0 && x0 < a.W, vx1 = x1 >= 0 && x1 < a.W, vy0 = y0 >= 0 && y0 < a.H, vy1 = y1 >= 0 && y1 < a.H;
  const float* base = a.fmap + (size_t)img * a.stride_img;
  for (int c = lane; c < a.C; c += 64) {
    const float* bc = base + (size_t)c * a.stride_c;
    const float v_nw = (vx0 && vy0) ? bc[(size_t)y0 * a.stride_h + (size_t)x0 * a.stride_w] : 0.f;
    const float v_ne = (vx1 && vy0) ? bc[(size_t)y0 * a.stride_h + (size_t)x1 * a.stride_w] : 0.f;
    const float v_sw = (vx0 && vy1) ? bc[(size_t)y1 * a.stride_h + (size_t)x0 * a.stride_w] : 0.f;
    const float v_se = (vx1 && vy1) ? bc[(size_t)y1 * a.stride_h + (size_t)x1 * a.stride_w] : 0.f;
    float acc;
    {
#pragma clang fp contract(off)
      acc = v_nw * w_nw;
    }
    acc = fmaf(v_ne, w_ne, acc);
    acc = fmaf(v_sw, w_sw, acc);
    acc = fmaf(v_se, w_se, acc);
    a.out[(size_t)p * a.C + c] = acc;
  }
}

__global__ void unpack_best_kernel(const unsigned long long* __restrict__ best, long long n, float* __restrict__ d2, int* __restrict__ idx) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned long long b = best[i];
  if (d2) d2[i] = __uint_as_float((unsigned)(b >> 32));
  idx[i] = (int)(b & 0xffffffffu);
}

__global__ void sqrt_inplace_kernel(float* x, long long n) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] = sqrtf(x[i]);
}

}  // namespace

int launch_sqnorm_rows(const float* x, long long n, int d, int ld, float* out, hipStream_t st) {
  FP_REQUIRE(d % 4 == 0 && ld % 4 == 0, "sqnorm_rows: d and ld must be multiples of 4");
  if (n == 0) return FP_OK;
  hipLaunchKernelGGL(sqnorm_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, n, d, ld, out);
  FP_CHECK_LAUNCH("sqnorm_rows");
  return FP_OK;
}

int launch_normalize_rows(const float* x, long long n, int d, float eps, float* out, hipStream_t st) {
  if (n == 0) return FP_OK;
  hipLaunchKernelGGL(normalize_rows_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, x, n, d, eps, out);
  FP_CHECK_LAUNCH("normalize_rows");
  return FP_OK;
}

int launch_topk_rows(const float* vals, int rows, int n, int ld, const int* row_len, int k, int largest,
                     float* out_val, int* out_idx, hipStream_t st) {
  FP_REQUIRE(k >= 1, "topk_rows: k must be >= 1");
  if (rows == 0) return FP_OK;
  hipLaunchKernelGGL(topk_rows_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, st, vals, rows, n, ld, row_len, k, largest, out_val, out_idx);
  FP_CHECK_LAUNCH("topk_rows");
  return FP_OK;
}

int launch_tfidf_build(const int* word_ids, const float* word_d2, int knn_k, const int* seg_off, int num_segs,
                       const float* idf, int num_words, int soft, float sigma_sq, int sqrt_dists,
                       float* desc, float* desc_n, float eps, hipStream_t st) {
  FP_REQUIRE(num_words > 0 && num_words <= 16384, "tfidf_build: num_words out of range");
  FP_REQUIRE(knn_k >= 1 && knn_k <= 16, "tfidf_build: knn_k out of range");
  if (num_segs == 0) return FP_OK;
  size_t lds = (size_t)num_words * 4 + 4096 * 8;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tfidf_build_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 16384 * 4 + 4096 * 8);
    attr = true;
  }
  hipLaunchKernelGGL(tfidf_build_kernel, dim3(num_segs), dim3(256), lds, st, word_ids, word_d2, knn_k, seg_off, idf,
                     num_words, soft, 2.0f * sigma_sq, sqrt_dists, desc, desc_n, eps);
  FP_CHECK_LAUNCH("tfidf_build");
  return FP_OK;
}

int launch_cyclic_select(const CyclicArgs& a, int num_pairs, hipStream_t st) {
  FP_REQUIRE(a.q_max <= 2048, "cyclic_select: more than 2048 query points per detection (got %d)", a.q_max);
  FP_REQUIRE(a.top_k >= 1 && a.k_max >= a.top_k, "cyclic_select: bad top_k / k_max");
  if (num_pairs == 0) return FP_OK;
  hipLaunchKernelGGL(cyclic_select_kernel, dim3(num_pairs), dim3(256), 0, st, a);
  FP_CHECK_LAUNCH("cyclic_select");
  return FP_OK;
}

int launch_sample_bilinear(const SampleArgs& a, hipStream_t st) {
  if (a.num_points == 0) return FP_OK;
  hipLaunchKernelGGL(sample_bilinear_kernel, dim3(cdiv(a.num_points, 4)), dim3(256), 0, st, a);
  FP_CHECK_LAUNCH("sample_bilinear");
  return FP_OK;
}

int launch_sqrt_inplace(float* x, long long n, hipStream_t st) {
  if (n == 0) return FP_OK;
  hipLaunchKernelGGL(sqrt_inplace_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, n);
  FP_CHECK_LAUNCH("sqrt_inplace");
  return FP_OK;
}

int launch_unpack_best(const unsigned long long* best, long long n, float* d2, int* idx, hipStream_t st) {
  if (n == 0) return FP_OK;
  hipLaunchKernelGGL(unpack_best_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, best, n, d2, idx);
  FP_CHECK_LAUNCH("unpack_best");
  return FP_OK;
}

int launch_topn_rows(float* sims, int ld, int rows, int max_len, const int* row_len, int n_top, float* out_scores,
                     int* out_ids, int tie_mode, int k_slices, long long slice_stride, unsigned long long* cand_scratch, hipStream_t st) {
  if (rows == 0) return FP_OK;
  if (tie_mode == 1) {
    FP_REQUIRE(max_len <= 20000, "strict (torch) tie order supports rows of at most 20000 elements (got %d)", max_len);
    const size_t lds = (size_t)max_len * 8;
    static size_t attr = 0;
    if (lds > attr) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&topn_rows_stl_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160000);
      attr = 160000;
    }
    hipLaunchKernelGGL(topn_rows_stl_kernel, dim3(rows), dim3(256), lds, st, sims, ld, row_len, max_len, n_top, out_scores, out_ids, k_slices, slice_stride);
  } else {
    FP_REQUIRE(n_top <= 8, "top-n: n_top must be <= 8 on the canonical block path");
    if (cand_scratch && max_len >= 4096) {
      const int nsplit = 128 / n_top < 24 ? 128 / n_top : 24;  // the merge wave holds nsplit * n_top <= 128 candidates
      hipLaunchKernelGGL(topn_rows_wave_kernel<8>, dim3(rows, cdiv(nsplit, 4)), dim3(256), 0, st, sims, ld, row_len, max_len, n_top,
                         k_slices, slice_stride, nsplit, rows, cand_scratch);
      hipLaunchKernelGGL(topn_merge_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, st, cand_scratch, nsplit * n_top, rows, n_top, out_scores, out_ids);
    } else {
      hipLaunchKernelGGL(topn_rows_block_kernel<8>, dim3(rows), dim3(256), 0, st, sims, ld, row_len, max_len, n_top, out_scores, out_ids,
                         k_slices, slice_stride, (unsigned long long*)nullptr);
    }
  }
  FP_CHECK_LAUNCH("topn_rows");
  return FP_OK;
}

int launch_cosine_topk(const CosineArgs& a, int num_det, int num_obj, int max_det_per_obj, int max_templates, int n_top,
                       const int* det_num_templates, float* out_scores, int* out_ids, int tie_mode, hipStream_t st) {
  FP_REQUIRE(a.W % 16 == 0, "cosine_topk: the streaming kernel needs num_words %% 16 == 0");
  FP_REQUIRE(max_det_per_obj <= 64, "cosine_topk: at most 64 detections per object per call (got %d); split the batch", max_det_per_obj);
  const int nq = cdiv(max_det_per_obj, 16) <= 1 ? 1 : (cdiv(max_det_per_obj, 16) == 2 ? 2 : 4);
  const int wslice = a.W / a.k_slices;
  const size_t lds = (size_t)nq * 16 * ((size_t)wslice * 4 + 16);
  FP_REQUIRE(lds <= 160 * 1024, "cosine_topk: query slice does not fit LDS (num_words %d)", a.W);
  static bool attr = false;
  static int num_cus = 256;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&cosine_sims_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&cosine_sims_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&cosine_sims_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&cosine_stream_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&cosine_stream_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) num_cus = n;
    attr = true;
  }
  // FP_COSINE_STREAM=0 (read once): A/B switch back to the register-direct kernel
  static const int env_stream = getenv("FP_COSINE_STREAM") ? atoi(getenv("FP_COSINE_STREAM")) : 1;
  if (env_stream && nq <= 2 && wslice % 64 == 0 && lds + 8 * 3 * 4096 <= 160 * 1024) {
    // one persistent workgroup per CU: the (object, slice) pairs share the CUs evenly.  8..12 waves per workgroup,
    // whichever splits the 16-template blocks most evenly over the waves (the kernel is close to MFMA-bound, so a
    // wave with 3 blocks next to waves with 2 costs what the slowest wave costs)
    const int nblk = cdiv(max_templates, 16);
    const int per = num_cus / (num_obj * a.k_slices);
    const int gx = per < 1 ? 1 : (per > cdiv(nblk, 8) ? cdiv(nblk, 8) : per);
    int nw = 8;
    double best = 0.0;
    for (int w = 8; w <= 12; ++w) {
      if (lds + (size_t)w * 3 * 4096 > 160 * 1024) break;
      const int slots = gx * w, mx = cdiv(nblk, slots);
      const double eff = (double)nblk / slots / mx;
      if (eff >= best) { best = eff; nw = w; }
    }
    const size_t lds_stream = lds + (size_t)nw * 3 * 4096;
    dim3 sgrid(gx, num_obj, a.k_slices);
    if (nq == 1) hipLaunchKernelGGL(cosine_stream_kernel<1>, sgrid, dim3(nw * 64), lds_stream, st, a);
    else hipLaunchKernelGGL(cosine_stream_kernel<2>, sgrid, dim3(nw * 64), lds_stream, st, a);
  } else {
    dim3 grid(cdiv(cdiv(max_templates, 16), 4), num_obj, a.k_slices);
    if (nq == 1) hipLaunchKernelGGL(cosine_sims_kernel<1>, grid, dim3(256), lds, st, a);
    else if (nq == 2) hipLaunchKernelGGL(cosine_sims_kernel<2>, grid, dim3(256), lds, st, a);
    else hipLaunchKernelGGL(cosine_sims_kernel<4>, grid, dim3(256), lds, st, a);
  }
  FP_CHECK_LAUNCH("cosine_sims");
  // candidate keys of the split top-n live behind the 8 slice buffers (scratch contract: 9 slices)
  unsigned long long* cand = reinterpret_cast<unsigned long long*>(a.sims + 8 * (size_t)num_det * a.ld_sims);
  return launch_topn_rows(a.sims, a.ld_sims, num_det, max_templates, det_num_templates, n_top, out_scores, out_ids, tie_mode,
                          a.k_slices, a.slice_stride, (24 * n_top * 3 <= a.ld_sims) ? cand : nullptr, st);
}
