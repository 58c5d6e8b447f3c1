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
#include "stl_wave.hpp"

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

// ------------------------------------------------------------------ template retrieval: cosine scores
// sims[det][t] = <bank_n[t,:], q_n[det,:]> for every template t of the detection's object, + top-n.
// v_mfma_f32_16x16x4_f32: A = 16 templates x 4 k, B = 4 k x 16 detections.  A lane's float4 covers
// k = 16j + 4g .. +3 (g = lane>>4), so MFMA step u consumes k = 16j + 4g' + u, g' = 0..3: the per-(template, detection)
// fp32 fma chain visits each 16-block of k in the order [0,4,8,12, 1,5,9,13, 2,6,10,14, 3,7,11,15].  K is cut into
// a.k_slices contiguous slices, each slice is one such chain starting from zero, and the slice sums are added in slice
// order: the canonical order of this stage (oracle: orc_dot_rows_perm16).  Both kernels below produce exactly that
// chain, so a score does not depend on which kernel ran or on how many detections share the launch.

// Generic shapes (any num_words % 16 == 0): 4 waves x 16 templates per workgroup, bank rows register-direct (16 B per
// lane), the query slice restaged in LDS for every k-slice.  Not tuned -- production shapes take cosine_fused_kernel.
template <int NQ>
__global__ __launch_bounds__(256) void cosine_generic_kernel(CosineArgs a) {
  extern __shared__ __attribute__((aligned(16))) char qlds[];  // [NQ*16 detections][wslice floats + 16 B pad]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int obj = blockIdx.y;
  const int tb = a.obj_tpl_off[obj], T = a.obj_tpl_off[obj + 1] - tb;
  const int d0 = a.det_seg_off[obj] + blockIdx.z * (NQ * 16);
  const int nd = min(NQ * 16, a.det_seg_off[obj + 1] - d0);
  if (blockIdx.x * 64 >= T || nd <= 0) return;  // block-uniform
  const int wslice = a.W / a.k_slices, pitch = wslice * 4 + 16;
  const int t0 = (blockIdx.x * 4 + wave) * 16;
  const int i = lane & 15, g = lane >> 4;
  const int trow = min(t0 + i, T - 1);
  const char* qs = qlds + i * pitch + g * 16;
  f32x4 tot[NQ];
  for (int sl = 0; sl < a.k_slices; ++sl) {
    __syncthreads();  // the previous slice has no readers left
    for (int r = wave; r < NQ * 16; r += 4) {
      const float* src = a.desc_n + (size_t)(d0 + min(r, nd - 1)) * a.W + sl * wslice;
      for (int c = lane * 4; c < wslice; c += 256)
        *reinterpret_cast<float4*>(qlds + r * pitch + c * 4) = *reinterpret_cast<const float4*>(src + c);
    }
    __syncthreads();
    const float* ap = a.bank_n + (size_t)(tb + trow) * a.W + sl * wslice + 4 * g;
    f32x4 acc[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int j = 0; j < wslice / 16; ++j) {
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
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      if (sl == 0) tot[q] = acc[q];
      else tot[q] += acc[q];  // slice sums added in slice order
    }
  }
  // D[i = template 4g + r][j = detection lane&15]
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int det = q * 16 + i;
    if (det >= nd || t0 >= T) continue;
    float* o = a.sims + (size_t)(d0 + det) * a.ld_sims + t0 + 4 * g;
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (t0 + 4 * g + r < T) o[r] = tot[q][r];
  }
}

// Production shape (num_words = 8 slices of 64..256 words): HBM-bound by design, one pass over the bank, no partial
// scores in memory.  One persistent 8-wave workgroup per CU; wave s owns k-slice s.  The <= 32 query descriptors of the
// launch live in REGISTERS (a wave holds its slice of all of them as MFMA B fragments: 128 VGPRs at 256 words x 32
// detections) -- LDS could hold only one slice of them, which is why the earlier version cut K across workgroups, wrote
// [8, B, T] partial scores and re-read them across XCDs in a second and third kernel.  The workgroup walks the object's
// 16-template blocks; the eight waves stream the same 16 rows (128 KiB contiguous per block), each its own 1-KiB
// segment of every row as 4-KiB chunks (16 rows x 256 B, whole row segments fetched by four global_load_lds, 16-B
// pieces XOR-placed by row so the fragment reads are conflict-free) through a private three-slot LDS ring with counted
// vmcnt waits; a landed chunk moves to registers at once so all three slots stay in flight (96 KiB per CU).  After a
// block's chunks the eight slice sums meet in LDS (one barrier per 128 KiB of bank), thread (detection, template) adds
// them in slice order, stores the finished score and keeps the best n_top it has seen as sorted (score, id) keys; at
// the end the 16 lanes of a detection merge their lists and the workgroup emits n_top candidate keys per detection.
typedef __attribute__((address_space(3))) void cos_lds_void;
typedef __attribute__((address_space(1))) const void cos_gbl_cvoid;
constexpr int COS_NMAX = 8;          // candidates kept per thread / emitted per (workgroup, detection)
constexpr int COS_RED_PITCH = 68;    // floats per 16-lane group of a wave's score tile (64 + 4: de-phases the groups across banks)
#ifndef FP_COS_SLOTS
#define FP_COS_SLOTS 3               // ring depth per wave, 4-KiB chunks (measurement builds: 2 / 4)
#endif
constexpr int COS_SLOTS = FP_COS_SLOTS;
constexpr int COS_RED_BUFS = COS_SLOTS <= 3 ? 2 : 1;  // a 4-slot ring leaves LDS for one reduction buffer (second barrier per block)
constexpr int COS_RING_BYTES = 8 * COS_SLOTS * 4096;

// BF = the approximate first pass of the prefiltered retrieval (fp_cosine_topk_prefiltered): the bank is its fp16 copy (half the
// bytes), the queries are rounded to fp16 on their way into the registers, one v_mfma_f32_16x16x32_f16 takes the place of four
// fp32 MFMAs (1/16 of the matrix time); the "scores" it leaves in a.sims are within COS_PREFILTER_EPS of the exact ones.
// (fp16, not bf16: rows are L2-normalised, so every element is <= 1 and the 11-bit mantissa gives a 4x tighter bound.)
template <int NQ, bool BF = false>
__global__ __launch_bounds__(512) void cosine_fused_kernel(CosineArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [8 waves][3][4 KiB] ring | red[2][8 slices][NQ][4][68] floats
  if (a.run_flag && *a.run_flag == 0) return;  // exact fallback of the prefiltered retrieval: runs only if some row asked for it
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int obj = blockIdx.y;
  const int tb = a.obj_tpl_off[obj], T = a.obj_tpl_off[obj + 1] - tb;
  const int d0 = a.det_seg_off[obj] + blockIdx.z * (NQ * 16);
  const int nd = min(NQ * 16, a.det_seg_off[obj + 1] - d0);
  if (nd <= 0) return;  // block-uniform: no detection rows, nothing to emit
  constexpr int ESZ = BF ? 2 : 4;                  // bytes per bank element
  const int wslice = a.W >> 3, nch = (wslice * ESZ) >> 8;  // 256-byte chunks (64 fp32 / 128 fp16 words) per slice (1..4)
  const int i = lane & 15, g = lane >> 4;
  char* ring = smem + wave * (COS_SLOTS * 4096);
  float* red = reinterpret_cast<float*>(smem + COS_RING_BYTES);
  constexpr int RED_SLICE = NQ * 4 * COS_RED_PITCH;  // floats per (buffer, slice)

  const int nblk = (T + 15) >> 4;
  const int first = blockIdx.x, stride = gridDim.x;
  const int ntask = first < nblk ? (nblk - first + stride - 1) / stride : 0;
  const int total = ntask * nch;

  // ---- this wave's slice of the query descriptors -> registers, in MFMA B-fragment order
  float4 qv[16][NQ];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        qv[c * 4 + j][q] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < nch && total > 0) {
          const float* qrow = a.desc_n + (size_t)(d0 + min(q * 16 + i, nd - 1)) * a.W + wave * wslice;
          if constexpr (BF) {  // piece (c*4+j): 8 consecutive words at 32 (c*4+j) + 8 g, rounded to fp16 (RNE)
            const float4 lo = *reinterpret_cast<const float4*>(qrow + (c * 4 + j) * 32 + g * 8);
            const float4 hi = *reinterpret_cast<const float4*>(qrow + (c * 4 + j) * 32 + g * 8 + 4);
            auto pk = [](float x, float y) { return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{x, y}, f16x2)); };
            qv[c * 4 + j][q] = __builtin_bit_cast(float4, make_uint4(pk(lo.x, lo.y), pk(lo.z, lo.w), pk(hi.x, hi.y), pk(hi.z, hi.w)));
          } else {
            qv[c * 4 + j][q] = *reinterpret_cast<const float4*>(qrow + (c * 4 + j) * 16 + g * 4);
          }
        }
      }

  // (byte addressing: a bank row is W * ESZ bytes, this wave's slice starts wave * wslice * ESZ bytes into it)
  const char* slice_base = reinterpret_cast<const char*>(BF ? a.bank_bf16 : (const void*)a.bank_n) + ((size_t)tb * a.W + (size_t)wave * wslice) * ESZ;
  int it_blk = first, it_ch = 0, it_slot = 0;
  auto issue = [&]() {
#ifdef FP_COS_NO_DMA
    return;
#endif
    const int t0 = it_blk * 16;
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      const int row = 4 * q4 + g;                     // lane -> (row of the block, physical 16-B slot i)
      const int piece = i ^ row;                      // logical piece that must land in slot i of this row
      const char* src = slice_base + (size_t)min(t0 + row, T - 1) * a.W * ESZ + it_ch * 256 + piece * 16;
      __builtin_amdgcn_global_load_lds((cos_gbl_cvoid*)src, (cos_lds_void*)(ring + it_slot * 4096 + q4 * 1024), 16, 0, 2 /* nt */);
    }
    if (++it_ch == nch) { it_ch = 0; it_blk += stride; }
    it_slot = it_slot == COS_SLOTS - 1 ? 0 : it_slot + 1;
  };
#pragma unroll
  for (int p = 0; p < COS_SLOTS; ++p)
    if (total > p) issue();

  // reduce-phase role of this thread: detection rd (of the launch's NQ*16), template rt of the block
  const int rd = tid >> 4, rt = tid & 15;
  const bool reducer = rd < NQ * 16;
  const int red_off = (rd >> 4) * (4 * COS_RED_PITCH) + (rt >> 2) * COS_RED_PITCH + (rd & 15) * 4 + (rt & 3);
  unsigned long long best[COS_NMAX];
#pragma unroll
  for (int s = 0; s < COS_NMAX; ++s) best[s] = ~0ull;

  f32x4 acc[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
  int blk = first, slot = 0, buf = 0;
  for (int task = 0; task < ntask; ++task) {
#pragma unroll
    for (int ch = 0; ch < 4; ++ch) {
      if (ch < nch) {
        const int cc = task * nch + ch;
        // loads return in order: chunk cc has landed once at most the later chunks' DMAs are outstanding
        const int ahead = total - 1 - cc < COS_SLOTS - 1 ? total - 1 - cc : COS_SLOTS - 1;  // chunks issued after this one
#ifndef FP_COS_NO_DMA  // (measurement builds: the kernel without its bank stream)
        if (ahead >= 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else if (ahead == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (ahead == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        const char* cs = ring + slot * 4096 + i * 256;
        f32x4 av[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) av[j] = *reinterpret_cast<const f32x4*>(cs + (((4 * j + g) ^ i) << 4));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // fragments are in registers before the slot is handed back
        if (cc + COS_SLOTS < total) issue();
#ifndef FP_COS_NO_MFMA  // (measurement builds, tools/cos_ablate.sh: the kernel without its matrix work)
        if constexpr (BF) {
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int q = 0; q < NQ; ++q)
              acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, av[j]), __builtin_bit_cast(f16x8, qv[ch * 4 + j][q]), acc[q], 0, 0, 0);
        } else
#pragma unroll
        for (int j = 0; j < 4; ++j) {  // two alternating accumulator chains (NQ = 2)
#pragma unroll
          for (int q = 0; q < NQ; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j][0], qv[ch * 4 + j][q].x, acc[q], 0, 0, 0);
#pragma unroll
          for (int q = 0; q < NQ; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j][1], qv[ch * 4 + j][q].y, acc[q], 0, 0, 0);
#pragma unroll
          for (int q = 0; q < NQ; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j][2], qv[ch * 4 + j][q].z, acc[q], 0, 0, 0);
#pragma unroll
          for (int q = 0; q < NQ; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j][3], qv[ch * 4 + j][q].w, acc[q], 0, 0, 0);
        }
#else
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int q = 0; q < NQ; ++q) acc[q] += av[j] * f32x4{qv[ch * 4 + j][q].x, qv[ch * 4 + j][q].y, qv[ch * 4 + j][q].z, qv[ch * 4 + j][q].w};
#endif
        slot = slot == COS_SLOTS - 1 ? 0 : slot + 1;
      }
    }
#ifdef FP_COS_NO_REDUCE  // (measurement builds: no slice reduction, no score store, no candidate lists)
    if (acc[0][0] == 1234.5f) a.sims[tid] = acc[0][1];
    blk += stride;
    continue;
#endif
    // ---- the block's eight slice sums meet in LDS: D[template 4g + r][detection q*16 + i] of this wave's slice
    float* mine = red + (buf * 8 + wave) * RED_SLICE;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      *reinterpret_cast<f32x4*>(mine + q * (4 * COS_RED_PITCH) + g * COS_RED_PITCH + i * 4) = acc[q];
      acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();  // (also orders this buffer's readers of two blocks ago before its next writers)
    if (reducer) {
      const float* rp = red + buf * 8 * RED_SLICE + red_off;
      float v = rp[0];
#pragma unroll
      for (int sl = 1; sl < 8; ++sl) v += rp[sl * RED_SLICE];  // slice sums added in slice order
      const int t = blk * 16 + rt;
      if (t < T && rd < nd) {
        a.sims[(size_t)(d0 + rd) * a.ld_sims + t] = v;
        // (a NaN of either sign ranks first, as in torch.topk: it then shows up among the candidates and sends the row to the replay)
        unsigned long long key = ((unsigned long long)order_key(v != v ? __uint_as_float(0x7fc00000u) : v, true) << 32) | (unsigned)t;
#pragma unroll
        for (int s = 0; s < COS_NMAX; ++s) {  // sorted insertion (ascending keys = best first)
          const unsigned long long lo = key < best[s] ? key : best[s];
          key = key < best[s] ? best[s] : key;
          best[s] = lo;
        }
      }
    }
    if (COS_RED_BUFS == 2) buf ^= 1;
    else __syncthreads();
    blk += stride;
  }
  // ---- the 16 lanes of a detection merge their lists: n_top candidate keys per (workgroup, detection)
  if (reducer && a.cand) {
    for (int s = 0; s < a.n_top; ++s) {
      unsigned long long m = best[0];
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) {
        const unsigned long long t = __shfl_xor(m, o, 16);
        m = t < m ? t : m;
      }
      if (m != ~0ull && best[0] == m) {  // exactly one lane owns the winner (keys carry the unique template id)
#pragma unroll
        for (int t = 0; t + 1 < COS_NMAX; ++t) best[t] = best[t + 1];
        best[COS_NMAX - 1] = ~0ull;
      }
      if (rt == 0 && rd < nd) a.cand[((size_t)(d0 + rd) * gridDim.x + blockIdx.x) * a.n_top + s] = m;
    }
  }
}

// ------------------------------------------------------------------ prefiltered retrieval: exact re-scoring of the candidates
// fp_cosine_topk_prefiltered = (1) cosine_fused_kernel<NQ, true>: approximate scores s~ from the fp16 bank and the best n + 1
// approximate keys of every workgroup.  Error bound for L2-normalised rows q, d (elements <= 1, W <= 4096 words): an element rounds
// to fp16 with |x~ - x| <= 2^-11 |x| + 2^-25 (normal range / subnormal spacing), so with Cauchy-Schwarz on unit rows
//   |sum q~ d~ - sum q d| <= 2 (2^-11 sum |q||d| + 2^-25 sqrt(W)) + (cross terms <= 2^-22 + ...) <= 2^-10 + 3.8e-6 + 2.4e-7 = 9.81e-4;
// the fp16 x fp16 products are exact in fp32; their accumulation (order and rounding inside the MFMA unspecified: one ulp = 2^-23
// of a running sum <= 1.001 per addition, W additions) adds <= 4.9e-4.  COS_PREFILTER_EPS = 2^-10 * 1.5625 = 1.526e-3 >= 1.47e-3.
// (2) this kernel: v = the (n + 1)-th best approximate score of the detection; every template of the exact top n + 1 has
// s~ >= v - 2 EPS (n + 1 templates have s >= v - EPS, so the exact (n + 1)-th best is >= v - EPS, and a template at or above it
// has s~ >= v - 2 EPS): those candidates -- however many -- get their EXACT score, computed with the fused kernel's own MFMA
// sequence (same instruction, same operand order, slice sums added in slice order => the same bits); (3) cosine_final_kernel:
// top n of the exact keys.  The strict (torch) tie order needs the whole row only when the best n + 1 exact scores contain a tie:
// those rows raise a flag, and the exact single-pass kernel + replay run behind it (they exit at once otherwise).
constexpr float COS_PREFILTER_EPS = 0.00152587890625f;  // 2^-10 * 1.5625
constexpr int COS_PARTS = 8;                         // a detection's template range is scanned by 8 workgroups
constexpr int COS_LIST_CAP = 8192;                   // candidates one workgroup can hold (a part is at most T / 8 templates: T <= 65536)

__global__ __launch_bounds__(512) void cosine_rescore_kernel(CosineArgs a, const unsigned long long* __restrict__ wg_keys, int keys_per_det, int n_emit,
                                                             const float* __restrict__ approx, unsigned long long* __restrict__ exact_keys,
                                                             int* __restrict__ exact_cnt, int key_stride) {
  __shared__ unsigned long long wmin[8];
  __shared__ int list[COS_LIST_CAP];
  __shared__ int cnt_s;
  __shared__ float red[8][16];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int part = blockIdx.x, det = blockIdx.y;
  int obj = 0;
  while (det >= a.det_seg_off[obj + 1]) ++obj;
  const int tb = a.obj_tpl_off[obj], T = a.obj_tpl_off[obj + 1] - tb;
  // ---- (a) the n_emit-th best approximate key of the detection (keys ascend with falling score)
  unsigned long long k[4];
  const unsigned long long* kp = wg_keys + (size_t)det * keys_per_det;
#pragma unroll
  for (int e = 0; e < 4; ++e) k[e] = tid + 512 * e < keys_per_det ? kp[tid + 512 * e] : ~0ull;
  unsigned long long nth = ~0ull;
  for (int r = 0; r < n_emit; ++r) {
    unsigned long long m = k[0] < k[1] ? k[0] : k[1];
    const unsigned long long m2 = k[2] < k[3] ? k[2] : k[3];
    m = m < m2 ? m : m2;
    m = wave_min_u64(m);
    if (lane == 0) wmin[wave] = m;
    __syncthreads();
    unsigned long long b = wmin[0];
#pragma unroll
    for (int w = 1; w < 8; ++w) b = wmin[w] < b ? wmin[w] : b;
    __syncthreads();
    nth = b;
    if (b == ~0ull) break;  // fewer than n_emit templates: everything is a candidate
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (k[e] == b) k[e] = ~0ull;  // (template ids are unique: exactly one holder)
  }
  float thr = -INFINITY;
  if (nth != ~0ull) {
    const unsigned kb = ~(unsigned)(nth >> 32);
    const float v = __uint_as_float((kb & 0x80000000u) ? (kb ^ 0x80000000u) : ~kb);
    thr = v - 2.f * COS_PREFILTER_EPS;  // NaN (a NaN score ranks first) -> no comparison below is true -> every template is re-scored
  }
  // ---- (b) this part's candidates
  if (tid == 0) cnt_s = 0;
  __syncthreads();
  const int tp = (T + COS_PARTS - 1) / COS_PARTS, t_lo = part * tp, t_hi = min(T, t_lo + tp);
  const float* arow = approx + (size_t)det * a.ld_sims;
  for (int t = t_lo + tid; t < t_hi; t += 512)
    if (!(arow[t] < thr)) list[atomicAdd(&cnt_s, 1)] = t;
  __syncthreads();
  const int cnt = cnt_s;
  // ---- (c) exact scores, 16 candidates at a time: wave s = k-slice s, the fused kernel's MFMA sequence on gathered rows
  const int wslice = a.W >> 3, i = lane & 15, g = lane >> 4;
  const float* qs = a.desc_n + (size_t)det * a.W + wave * wslice + 4 * g;
  unsigned long long* out = exact_keys + ((size_t)det * COS_PARTS + part) * key_stride;  // (key_stride >= every object's part length)
  for (int c0 = 0; c0 < cnt; c0 += 16) {
    const int trow = list[min(c0 + i, cnt - 1)];
    const float* ap = a.bank_n + (size_t)(tb + trow) * a.W + wave * wslice + 4 * g;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int j0 = 0; j0 < wslice / 16; j0 += 8) {  // eight 16-word steps per batch: all 16 loads in flight before the chain consumes them
      f32x4 av[8];
      float4 bv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        av[u] = *reinterpret_cast<const f32x4*>(ap + 16 * (j0 + u));
        bv[u] = *reinterpret_cast<const float4*>(qs + 16 * (j0 + u));
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][0], bv[u].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][1], bv[u].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][2], bv[u].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][3], bv[u].w, acc, 0, 0, 0);
      }
    }
    // D[template 4g + r][column lane & 15]; every column holds the same detection: column 0 reports
    if (i == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) red[wave][4 * g + r] = acc[r];
    }
    __syncthreads();
    if (tid < 16 && c0 + tid < cnt) {
      float v = red[0][tid];
#pragma unroll
      for (int sl = 1; sl < 8; ++sl) v += red[sl][tid];  // slice sums added in slice order
      out[c0 + tid] = ((unsigned long long)order_key(v != v ? __uint_as_float(0x7fc00000u) : v, true) << 32) | (unsigned)list[c0 + tid];
    }
    __syncthreads();
  }
  if (tid == 0) exact_cnt[det * COS_PARTS + part] = cnt;
}

// Top n of a detection's exact candidate keys (one wave per detection), the tie test of cand_merge_kernel, and the flag that
// releases the exact single-pass fallback when a row of the strict (torch) order has a tie among its best n + 1 scores.
__global__ __launch_bounds__(256) void cosine_final_kernel(const unsigned long long* __restrict__ exact_keys, const int* __restrict__ exact_cnt, int rows, int n_top,
                                                           float* __restrict__ out_val, int* __restrict__ out_idx, int* __restrict__ need_replay,
                                                           int* __restrict__ any_flag, int key_stride) {
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  unsigned long long best[COS_NMAX];
#pragma unroll
  for (int s = 0; s < COS_NMAX; ++s) best[s] = ~0ull;
  // lane = (part, j): the first 8 keys of every part in ONE round of loads (a part rarely holds more); longer parts loop
  const int p = lane >> 3, j0 = lane & 7;
  const unsigned long long* c = exact_keys + ((size_t)row * COS_PARTS + p) * key_stride;
  const int n = exact_cnt[row * COS_PARTS + p];
  for (int j = j0; j < n; j += 8) {
    unsigned long long key = c[j];
#pragma unroll
    for (int s = 0; s < COS_NMAX; ++s) {
      const unsigned long long lo = key < best[s] ? key : best[s];
      key = key < best[s] ? best[s] : key;
      best[s] = lo;
    }
  }
  float prev = 0.f;
  int tie = 0;
  const int rounds = need_replay ? n_top + 1 : n_top;
  for (int s = 0; s < rounds; ++s) {
    const unsigned long long b = wave_min_u64(best[0]);
    if (b == ~0ull) {
      if (lane == 0 && s < n_top) { out_idx[(size_t)row * n_top + s] = -1; out_val[(size_t)row * n_top + s] = -INFINITY; }
      continue;
    }
    const unsigned kb = ~(unsigned)(b >> 32);
    const float val = __uint_as_float((kb & 0x80000000u) ? (kb ^ 0x80000000u) : ~kb);
    if (val != val || (s > 0 && !(prev > val))) tie = 1;  // wave-uniform
    prev = val;
    if (best[0] == b) {
      if (s < n_top) {
        out_idx[(size_t)row * n_top + s] = (int)(b & 0xffffffffu);
        out_val[(size_t)row * n_top + s] = val;
      }
#pragma unroll
      for (int t = 0; t + 1 < COS_NMAX; ++t) best[t] = best[t + 1];
      best[COS_NMAX - 1] = ~0ull;
    }
  }
  if (need_replay && lane == 0) {
    need_replay[row] = tie;
    if (tie) atomicOr(any_flag, 1);
  }
}

// Canonical top-n of each row from the candidate keys of cosine_fused_kernel: one wave per detection, per-lane sorted
// lists over a strided share of the ncand keys, then n rounds of wave-wide arg-best.  The score travels inside the key.
// need_replay (torch tie order): the workgroups emitted n_top + 1 candidates; if the best n_top + 1 scores of the row are
// strictly decreasing, the top-n SET and its ORDER are unique, so every correct top-k -- torch.topk's partial_sort /
// nth_element + sort included -- returns exactly this list and the row's replay is skipped (flag 0).  Any equal pair, a
// +-0 pair or a NaN among them sets the flag and topn_rows_strict_kernel redoes the row from the scores.
FP_DEVICE void cand_merge_row(const unsigned long long* __restrict__ cand, int ncand, int n_top, float* __restrict__ out_val, int* __restrict__ out_idx,
                              int* __restrict__ need_replay, int row, int* tie_out) {
  // One 256-thread block per detection (a wave per detection walked the keys in six dependent rounds of loads: 8 us of latency):
  // every thread takes its keys in ONE round of loads, each wave extracts its best n + 1 with wave-wide minima, wave 0 merges the four lists.
  __shared__ unsigned long long wbest[4][COS_NMAX];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned long long* c = cand + (size_t)row * ncand;
  unsigned long long k[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) k[e] = tid + 256 * e < ncand ? c[tid + 256 * e] : ~0ull;
  for (int j = tid + 2048; j < ncand; j += 256) {  // (more than 2048 keys: not a shape the launcher produces; fold the rest in)
    const unsigned long long key = c[j];
    int worst = 0;
#pragma unroll
    for (int e = 1; e < 8; ++e) worst = k[e] > k[worst] ? e : worst;
    if (key < k[worst]) k[worst] = key;
  }
  const int rounds = need_replay ? n_top + 1 : n_top;  // <= COS_NMAX
  for (int s = 0; s < rounds; ++s) {
    unsigned long long m = k[0];
#pragma unroll
    for (int e = 1; e < 8; ++e) m = k[e] < m ? k[e] : m;
    const unsigned long long b = wave_min_u64(m);
    if (lane == 0) wbest[wave][s] = b;
    if (b != ~0ull) {
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (k[e] == b) k[e] = ~0ull;  // keys carry the unique template id: exactly one holder
    }
  }
  __syncthreads();
  if (wave != 0) return;
  unsigned long long mine = lane < 4 * rounds ? wbest[lane / rounds][lane % rounds] : ~0ull;
  float prev = 0.f;
  int tie = 0;
  for (int s = 0; s < rounds; ++s) {
    const unsigned long long b = wave_min_u64(mine);
    if (b == ~0ull) {
      if (lane == 0 && s < n_top) { out_idx[(size_t)row * n_top + s] = -1; out_val[(size_t)row * n_top + s] = -INFINITY; }
      continue;
    }
    const unsigned kb = ~(unsigned)(b >> 32);  // order_key inverted: the score's own bits
    const float val = __uint_as_float((kb & 0x80000000u) ? (kb ^ 0x80000000u) : ~kb);
    if (val != val || (s > 0 && !(prev > val))) tie = 1;  // wave-uniform
    prev = val;
    if (mine == b) {
      if (s < n_top) {
        out_idx[(size_t)row * n_top + s] = (int)(b & 0xffffffffu);
        out_val[(size_t)row * n_top + s] = val;
      }
      mine = ~0ull;
    }
  }
  if (need_replay && lane == 0) need_replay[row] = tie;
  if (tie_out && need_replay && lane == 0) *tie_out = tie;
}

__global__ __launch_bounds__(256) void cand_merge_kernel(const unsigned long long* __restrict__ cand, int ncand, int rows, int n_top,
                                                         float* __restrict__ out_val, int* __restrict__ out_idx, int* __restrict__ need_replay) {
  cand_merge_row(cand, ncand, n_top, out_val, out_idx, need_replay, blockIdx.x, nullptr);
}

// Canonical top-n of each row (largest first, ties -> lowest index) straight from the scores, one 256-thread block per
// row: per-thread top-n over a strided slice (registers), then n rounds of block-wide arg-best over the candidates.
template <int NMAX>
__global__ __launch_bounds__(256) void topn_rows_block_kernel(const float* __restrict__ vals, int ld, const int* __restrict__ row_len,
                                                              int n_default, int n_top, float* __restrict__ out_val, int* __restrict__ out_idx) {
  __shared__ unsigned long long cand[256 * NMAX];
  __shared__ unsigned long long wbest[4];
  const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int len = row_len ? row_len[row] : n_default;
  const float* r = vals + (size_t)row * ld;
  unsigned long long best[NMAX];
#pragma unroll
  for (int s = 0; s < NMAX; ++s) best[s] = ~0ull;
  for (int j = tid; j < len; j += 256) {
    unsigned long long key = ((unsigned long long)order_key(r[j], true) << 32) | (unsigned)j;
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
    if (tid == 0) {
      if (b != ~0ull) {
        const int j = (int)(b & 0xffffffffu);
        out_idx[(size_t)row * n_top + s] = j;
        out_val[(size_t)row * n_top + s] = r[j];
      } else {
        out_idx[(size_t)row * n_top + s] = -1;
        out_val[(size_t)row * n_top + s] = -INFINITY;
      }
    }
    prev = b;
    __syncthreads();
  }
}

// Strict-order top-n: the reference's torch.topk(scores, n) on a CPU tensor, ties included (stl_order.hpp), one block per
// row, rows of any length.  ATen runs std::partial_sort when n*64 <= len: a heap of the n best seen so far, and an
// element only acts when it beats the heap's root -- a handful of times in a row of thousands, most of them early.
//   phase 1  wave 0 replays the first STRICT_HEAD elements exactly: 64 lanes test four 64-element chunks against the
//            current root at once, only the hits go through the sequential pop_heap, in index order, the root re-read
//            after each.  The heap lives in registers, element j in lane j (LaneHeap).
//   phase 2  the root only ever improves, so an element that does not beat the root r1 left by phase 1 can never act.
//            All four waves scan the rest of the row straight from memory (each a contiguous quarter, 16-byte loads,
//            sixteen in flight) and keep, in index order, the few elements that beat r1.
//   phase 3  wave 0 replays those candidates like phase 1.
// Element moves inside the heap are libstdc++'s, so the surviving order among ties is too.  A row that overflows a
// candidate list (scores ascending along the row: every element acts) is replayed from memory chunk by chunk instead.
// Short rows (n*64 > len) take nth_element + sort on one lane, as ATen does.
constexpr int STRICT_HEAD = 2048;   // elements replayed in phase 1 (also the Elem capacity of the short-row branch / 2)
constexpr int STRICT_CAND = 1024;   // candidate capacity per wave

// The n-element heap of the replay, element j in the registers of lane j: reading heap[j] is a v_readlane, writing it a
// predicated move -- a pop_heap costs ~100 cycles instead of the ~1000 of dependent LDS round trips.
struct LaneHeap {
  float v;
  int idx;
  int lane;
  __device__ __forceinline__ stl_order::Elem get(int i) const {
    return stl_order::Elem{__int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), i)), __builtin_amdgcn_readlane(idx, i)};
  }
  __device__ __forceinline__ void set(int i, const stl_order::Elem& e) {
    if (lane == i) { v = e.v; idx = e.idx; }
  }
};

// 64 (value, index) pairs, one per lane in index order, against the heap: the partial_sort inner loop for these elements.
__device__ __forceinline__ void strict_replay64(LaneHeap& heap, int k, float v, int idx, bool ok, int lane) {
  stl_order::Elem top = heap.get(0);
  unsigned long long me = __ballot(ok && stl_order::gt(stl_order::Elem{v, 0}, top));
  while (me) {
    const int l = __builtin_ctzll(me);
    const stl_order::Elem x{__int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)), __builtin_amdgcn_readlane(idx, l)};
    stl_order::adjust_heap_acc(heap, 0, k, x);  // __pop_heap(first, middle, i): the old root leaves, *i enters
    top = heap.get(0);
    me = __ballot(ok && lane > l && stl_order::gt(stl_order::Elem{v, 0}, top));
  }
}

// One row's top n in torch.topk's order, replayed from the row of scores by a whole 256-thread block (block-uniform call).
FP_DEVICE void topn_row_strict(const float* __restrict__ vals, int ld, const int* __restrict__ row_len, int n_default, int n_top,
                               float* __restrict__ out_val, int* __restrict__ out_idx, int row) {
  __shared__ __attribute__((aligned(16))) float head[2 * STRICT_HEAD];   // phase 1 staging; the short-row branch's (value, index) pairs
  __shared__ float cand_v[4][STRICT_CAND];
  __shared__ int cand_i[4][STRICT_CAND];
  __shared__ int cand_n[4];
  __shared__ float s_root_v;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int len = row_len ? row_len[row] : n_default;
  const float* r = vals + (size_t)row * ld;
  const int k = min(n_top, len);
  if ((long long)k * 64 > (long long)len) {  // len < 64 * n_top: at most 4096 elements... capped to the LDS array below
    stl_order::Elem* el = reinterpret_cast<stl_order::Elem*>(head);
    for (int j = tid; j < len; j += 256) el[j] = stl_order::Elem{r[j], j};
    __syncthreads();
    if (tid == 0) stl_order::topk_torch_largest(el, len, k);
    __syncthreads();
    if (tid < n_top) {
      out_idx[(size_t)row * n_top + tid] = tid < k ? el[tid].idx : -1;
      out_val[(size_t)row * n_top + tid] = tid < k ? el[tid].v : -INFINITY;
    }
    return;
  }
  // ---- phase 1
  const int p1 = min(len, STRICT_HEAD);
  for (int j = tid; j < p1; j += 256) head[j] = r[j];
  if (tid < 4) cand_n[tid] = 0;
  __syncthreads();
  LaneHeap heap{0.f, 0, lane};
  if (wave == 0) {
    if (lane < k) { heap.v = head[lane]; heap.idx = lane; }
    stl_order::make_heap_acc(heap, k);  // std::make_heap over the first k elements (k <= 64 = one per lane)
    for (int c0 = k; c0 < p1; c0 += 64) {
      const int j = c0 + lane;
      strict_replay64(heap, k, j < p1 ? head[j] : 0.f, j, j < p1, lane);
    }
    if (lane == 0) s_root_v = heap.v;
  }
  __syncthreads();
  // ---- phase 2: each wave filters a contiguous quarter of [p1, len) against the root of phase 1
  const stl_order::Elem r1{s_root_v, 0};
  const int rest = len - p1, per = ((rest + 3) / 4 + 255) & ~255;  // quarter length, a multiple of 256
  const int q0 = p1 + wave * per, q1 = min(len, q0 + per);
  const bool vec = (ld & 3) == 0 && (p1 & 3) == 0;
  bool overflow = false;
  int n_c = 0;
  for (int base = q0; base < q1; base += 16 * 256) {
    float4 x[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int j = base + u * 256 + lane * 4;
      x[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (j + 4 <= q1 && vec) x[u] = *reinterpret_cast<const float4*>(r + j);
      else if (j < q1) {
        x[u].x = r[j];
        if (j + 1 < q1) x[u].y = r[j + 1];
        if (j + 2 < q1) x[u].z = r[j + 2];
        if (j + 3 < q1) x[u].w = r[j + 3];
      }
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int j = base + u * 256 + lane * 4;
      const float xe[4] = {x[u].x, x[u].y, x[u].z, x[u].w};
      unsigned long long bal[4];
      bool f[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        f[e] = j + e < q1 && stl_order::gt(stl_order::Elem{xe[e], 0}, r1);
        bal[e] = __ballot(f[e]);
      }
      if ((bal[0] | bal[1] | bal[2] | bal[3]) == 0) continue;  // wave-uniform: the common case
      const unsigned long long below = (1ull << lane) - 1ull;
      int pos = n_c + __popcll(bal[0] & below) + __popcll(bal[1] & below) + __popcll(bal[2] & below) + __popcll(bal[3] & below);
      const int total = __popcll(bal[0]) + __popcll(bal[1]) + __popcll(bal[2]) + __popcll(bal[3]);
      if (n_c + total > STRICT_CAND) { overflow = true; break; }
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (f[e]) { cand_v[wave][pos] = xe[e]; cand_i[wave][pos] = j + e; ++pos; }
      n_c += total;
    }
    if (overflow) break;
  }
  if (lane == 0) cand_n[wave] = overflow ? -1 : n_c;
  __syncthreads();
  // ---- phase 3
  if (wave == 0) {
    const bool any_overflow = cand_n[0] < 0 || cand_n[1] < 0 || cand_n[2] < 0 || cand_n[3] < 0;
    if (any_overflow) {  // adversarial order: plain chunked replay of the rest of the row from memory
      for (int c0 = p1; c0 < len; c0 += 64) {
        const int j = c0 + lane;
        strict_replay64(heap, k, j < len ? r[j] : 0.f, j, j < len, lane);
      }
    } else {
      for (int w = 0; w < 4; ++w)
        for (int c0 = 0; c0 < cand_n[w]; c0 += 64) {
          const int c = c0 + lane;
          const bool ok = c < cand_n[w];
          strict_replay64(heap, k, ok ? cand_v[w][c] : 0.f, ok ? cand_i[w][c] : 0, ok, lane);
        }
    }
    stl_order::sort_heap_acc(heap, k);
    if (lane < n_top) {
      out_idx[(size_t)row * n_top + lane] = lane < k ? heap.idx : -1;
      out_val[(size_t)row * n_top + lane] = lane < k ? heap.v : -INFINITY;
    }
  }
}

__global__ __launch_bounds__(256) void topn_rows_strict_kernel(const float* __restrict__ vals, int ld, const int* __restrict__ row_len,
                                                               int n_default, int n_top, float* __restrict__ out_val, int* __restrict__ out_idx,
                                                               const int* __restrict__ need_replay) {
  const int row = blockIdx.x;
  if (need_replay && !need_replay[row]) return;  // the row's top n + 1 scores are distinct: cand_merge_kernel's list is the answer
  topn_row_strict(vals, ld, row_len, n_default, n_top, out_val, out_idx, row);
}

// cand_merge_kernel + the replay of the rows it flags, one launch: a row whose best n + 1 candidate scores tie is redone from its scores by
// the same block (the two dependent launches were 7.5 + 5 us of a 37-us call).
__global__ __launch_bounds__(256) void cand_merge_replay_kernel(const unsigned long long* __restrict__ cand, int ncand, int rows, int n_top,
                                                                float* __restrict__ out_val, int* __restrict__ out_idx, int* __restrict__ need_replay,
                                                                const float* __restrict__ vals, int ld, const int* __restrict__ row_len, int n_default) {
  __shared__ int s_tie;
  const int row = blockIdx.x;
  if (threadIdx.x == 0) s_tie = need_replay ? 0 : 1;  // no flag array: every row is replayed, as with the two launches
  __syncthreads();
  cand_merge_row(cand, ncand, n_top, out_val, out_idx, need_replay, row, &s_tie);
  __syncthreads();
  if (s_tie) topn_row_strict(vals, ld, row_len, n_default, n_top, out_val, out_idx, row);  // (block-uniform) overwrites the merged list
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
  const float fQ = (float)Q;
  const int total = Q * knn_k;
  if (!soft) {
    // Hard assignment (the shipped options): every entry of a bin adds the SAME value tf * idf[bin] (weights are all 1, so
    // tf = (1 / sqrt(k)) / Q for every entry), and a run of n equal fp32 addends sums to the same bits in any order.  So the
    // order-sensitive walk below is not needed: count the entries per bin with integer LDS atomics (exact, order-free,
    // all 256 threads busy), then each bin replays its n sequential additions -- bit-identical to scatter_add_, ~10x faster.
    int* cnt = reinterpret_cast<int*>(bins);
    for (int w = tid; w < num_words; w += 256) cnt[w] = 0;
    __syncthreads();
    for (int e = tid; e < total; e += 256) {
      const int id = word_ids[(size_t)q0 * knn_k + e];
      if (id >= 0) atomicAdd(&cnt[id], 1);
    }
    __syncthreads();
    float nrm2 = 0.f;
    for (int t = 0; t < knn_k; ++t) nrm2 = nrm2 + 1.f;  // sequential, like the general path below
    const float tf = (1.f / fmaxf(sqrtf(nrm2), 1e-12f)) / fQ;
    for (int w = tid; w < num_words; w += 256) {
      const int n = cnt[w];
      const float v = tf * idf[w];
      float acc = 0.f;
      for (int i = 0; i < n; ++i) acc += v;
      bins[w] = acc;  // same LDS word, read as int above by this thread only
    }
  } else {
  for (int w = tid; w < num_words; w += 256) bins[w] = 0.f;
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
        float x = sqrt_dists ? sqrtf(dr[t]) : dr[t];
        const float w = expf(-(x * x) / two_sigma_sq);
        nrm2 = nrm2 + w * w;  // sequential, like a 3-element fp32 sum
        if (t == j) wj = w;
      }
      const float wn = wj / fmaxf(sqrtf(nrm2), 1e-12f);
      const float tf = wn / fQ;
      const int id = idr[j];  // -1: the k-NN had fewer than knn_k words to offer (padding, like faiss) -> no bin
      ids[e] = id;
      vals[e] = id >= 0 ? tf * idf[id] : 0.f;
    }
    __syncthreads();
    for (int e = 0; e < cnt; ++e) {
      const int id = ids[e];
      if (id >= 0 && (id & 255) == tid) bins[id] += vals[e];
    }
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
  __shared__ unsigned short lpos_s[2048], rpos_s[2048];  // strict mode: stopper ranks of the wave-parallel partition
  __shared__ stl_order::Elem tmp_s[2048];                // strict mode: target of the final stable placement
  const int pair = blockIdx.x, tid = threadIdx.x;
  const int det = pair / a.n_slots;
  const int q0 = a.q_off[det], Q = a.q_off[det + 1] - q0;
  int tpl = a.tpl_ids[pair];
  if (tpl >= 0 && a.tpl_base) tpl += a.tpl_base[det];
  const int kk = min(a.top_k, Q);
  // an empty slot (fewer templates than n_slots) or a template without features yields no correspondences
  const bool live = tpl >= 0 && a.tpl_off[tpl + 1] > a.tpl_off[tpl] && Q > 0;
  if (tid == 0) a.out_count[pair] = live ? kk : 0;
  // the kernel owns the whole padded record: entries past the count are written here (-1 ids, zeros), so the caller hands
  // over uninitialised buffers (seven fill kernels per batch otherwise)
  for (int r = (live ? kk : 0) + tid; r < a.k_max; r += 256) {
    const size_t o = (size_t)pair * a.k_max + r;
    a.out_q_ids[o] = -1;
    a.out_feat_ids[o] = -1;
    a.out_dists[o] = 0.f;
    a.out_conf[o] = 0.f;
    a.out_coord_2d[o * 2] = 0.f; a.out_coord_2d[o * 2 + 1] = 0.f;
    a.out_coord_3d[o * 3] = 0.f; a.out_coord_3d[o * 3 + 1] = 0.f; a.out_coord_3d[o * 3 + 2] = 0.f;
  }
  if (!live) return;
  const int f0 = a.tpl_off[tpl];
  // the distance tiles left one slice of nearest-neighbour keys per tile: the nearest over the whole template / crop is the
  // smallest key over the live tiles (keys order by distance, then index: the same winner an atomicMin would have kept)
  // (row_parts / col_parts == 1: the two-stage search of knn_cand.hip left ONE finished key per row / column)
  const int np = a.row_parts == 1 ? 1 : (a.tpl_off[tpl + 1] - f0 + 127) / 128, nq = a.col_parts == 1 ? 1 : (Q + 127) / 128;
  const unsigned long long* rb = a.row_best + (size_t)pair * a.row_parts * a.row_stride;
  const unsigned long long* cb = a.col_best + (size_t)pair * a.col_parts * a.col_stride;
  const float* pts = a.points + (size_t)q0 * 2;

  for (int i = tid; i < 2048; i += 256) {
    unsigned long long key = ~0ull;
    if (i < Q) {
      unsigned long long kr = rb[i];
      for (int t = 1; t < np; ++t) {
        const unsigned long long v = rb[(size_t)t * a.row_stride + i];
        kr = v < kr ? v : kr;
      }
      const int o = (int)(kr & 0xffffffffu);           // query -> nearest template patch
      unsigned long long kc = cb[o];
      for (int t = 1; t < nq; ++t) {
        const unsigned long long v = cb[(size_t)t * a.col_stride + o];
        kc = v < kc ? v : kc;
      }
      const int c = (int)(kc & 0xffffffffu);           // that patch -> nearest query patch
      q2o_s[i] = o;
      key = pack_dist_idx(point_dist(pts[2 * i], pts[2 * i + 1], pts[2 * c], pts[2 * c + 1]), (unsigned)i);
    }
    keys[i] = key;
  }
  __syncthreads();
  if (a.tie_mode == 1) {
    // strict mode: the reference's torch.topk(-cycle_dists, k) order, ties included -- one wave replays
    // libstdc++'s nth_element + sort (or partial_sort) on (value, index) pairs held in LDS (stl_wave.hpp)
    stl_order::Elem* el = reinterpret_cast<stl_order::Elem*>(keys);
    for (int i = tid; i < Q; i += 256) {
      const unsigned long long key = keys[i];  // each thread converts only the slots it reads itself
      el[i] = stl_order::Elem{-__uint_as_float((unsigned)(key >> 32)), i};
    }
    __syncthreads();
    if (tid < 64) stl_wave::topk_torch_largest(el, Q, kk, lpos_s, rpos_s, tmp_s, tid);
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

// Finishes the fused k-NN: per row the k best of its ncand = n_tiles * k candidate keys (canonical: smallest (d2, index)).
// 16 lanes per row: per-lane sorted lists over a strided share, then k rounds of 16-lane arg-min + pop.
__global__ __launch_bounds__(256) void knn_merge_kernel(const unsigned long long* __restrict__ cand, int rows, int ncand, int k,
                                                        float* __restrict__ out_d2, int* __restrict__ out_idx) {
  const int row = blockIdx.x * 16 + (threadIdx.x >> 4), l = threadIdx.x & 15;
  const bool live = row < rows;
  constexpr int KMAX = 8;
  unsigned long long best[KMAX];
#pragma unroll
  for (int s = 0; s < KMAX; ++s) best[s] = ~0ull;
  if (live)
    for (int c = l; c < ncand; c += 16) {
      unsigned long long key = cand[(size_t)row * ncand + c];
#pragma unroll
      for (int s = 0; s < KMAX; ++s) {
        const unsigned long long lo = key < best[s] ? key : best[s];
        key = key < best[s] ? best[s] : key;
        best[s] = lo;
      }
    }
  for (int s = 0; s < k; ++s) {
    unsigned long long m = best[0];
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {
      const unsigned long long t = __shfl_xor(m, o, 16);
      m = t < m ? t : m;
    }
    if (m != ~0ull && best[0] == m) {
#pragma unroll
      for (int t = 0; t + 1 < KMAX; ++t) best[t] = best[t + 1];
      best[KMAX - 1] = ~0ull;
    }
    if (live && l == 0) {
      out_idx[(size_t)row * k + s] = m == ~0ull ? -1 : (int)(m & 0xffffffffu);
      out_d2[(size_t)row * k + s] = m == ~0ull ? INFINITY : __uint_as_float((unsigned)(m >> 32));
    }
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

namespace {
// The fixed-size record of a detection for the final gather (engine.pack_result): n x (template id, score, count) then n x K x
// (query id, feature id, distance, confidence, x, y, X, Y, Z), every field one 32-bit word; integers keep their bit patterns.
__global__ __launch_bounds__(256) void pack_records_kernel(const int* __restrict__ tpl_ids, const float* __restrict__ scores, const int* __restrict__ counts,
                                                           const int* __restrict__ q_ids, const int* __restrict__ feat_ids, const float* __restrict__ dists,
                                                           const float* __restrict__ conf, const float* __restrict__ c2d, const float* __restrict__ c3d,
                                                           int n, int K, unsigned* __restrict__ out) {
  const int pair = blockIdx.x;  // (detection, slot)
  unsigned* rec = out + (size_t)pair * (3 + (size_t)K * 9);
  if (threadIdx.x == 0) {
    rec[0] = (unsigned)tpl_ids[pair];
    rec[1] = __float_as_uint(scores[pair]);
    rec[2] = (unsigned)counts[pair];
  }
  for (int k = threadIdx.x; k < K; k += 256) {
    const size_t o = (size_t)pair * K + k;
    unsigned* r = rec + 3 + (size_t)k * 9;
    r[0] = (unsigned)q_ids[o];
    r[1] = (unsigned)feat_ids[o];
    r[2] = __float_as_uint(dists[o]);
    r[3] = __float_as_uint(conf[o]);
    r[4] = __float_as_uint(c2d[o * 2]);
    r[5] = __float_as_uint(c2d[o * 2 + 1]);
    r[6] = __float_as_uint(c3d[o * 3]);
    r[7] = __float_as_uint(c3d[o * 3 + 1]);
    r[8] = __float_as_uint(c3d[o * 3 + 2]);
  }
}
}  // namespace

int launch_pack_records(const int* tpl_ids, const float* scores, const int* counts, const int* q_ids, const int* feat_ids, const float* dists, const float* conf,
                        const float* c2d, const float* c3d, int num_det, int n, int K, float* out, hipStream_t st) {
  if (num_det * n == 0) return FP_OK;
  hipLaunchKernelGGL(pack_records_kernel, dim3(num_det * n), dim3(256), 0, st, tpl_ids, scores, counts, q_ids, feat_ids, dists, conf, c2d, c3d, n, K,
                     reinterpret_cast<unsigned*>(out));
  FP_CHECK_LAUNCH("pack_records");
  return FP_OK;
}

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
  static FpDeviceOnce attr;
  fp_allow_dynamic_lds(attr, &tfidf_build_kernel, 16384 * 4 + 4096 * 8);
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

int launch_knn_merge(const unsigned long long* cand, int rows, int ncand, int k, float* out_d2, int* out_idx, hipStream_t st) {
  FP_REQUIRE(k >= 1 && k <= 8, "knn_merge: k must be in [1, 8]");
  if (rows == 0) return FP_OK;
  hipLaunchKernelGGL(knn_merge_kernel, dim3(cdiv(rows, 16)), dim3(256), 0, st, cand, rows, ncand, k, out_d2, out_idx);
  FP_CHECK_LAUNCH("knn_merge");
  return FP_OK;
}

int launch_unpack_best(const unsigned long long* best, long long n, float* d2, int* idx, hipStream_t st) {
  if (n == 0) return FP_OK;
  hipLaunchKernelGGL(unpack_best_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, best, n, d2, idx);
  FP_CHECK_LAUNCH("unpack_best");
  return FP_OK;
}

// Top-n of finished scores [rows, ld]: tie_mode 1 = the reference's torch.topk order, 0 = canonical.
int launch_topn_rows(const float* sims, int ld, int rows, int max_len, const int* row_len, int n_top, float* out_scores,
                     int* out_ids, int tie_mode, hipStream_t st, const int* need_replay) {
  if (rows == 0) return FP_OK;
  if (tie_mode == 1) {
    FP_REQUIRE(n_top <= 32, "strict (torch) tie order: n_top must be <= 32 (got %d)", n_top);  // short rows (< 64 n_top) fit the LDS pair array
    hipLaunchKernelGGL(topn_rows_strict_kernel, dim3(rows), dim3(256), 0, st, sims, ld, row_len, max_len, n_top, out_scores, out_ids, need_replay);
  } else if (n_top <= 8) {
    hipLaunchKernelGGL(topn_rows_block_kernel<8>, dim3(rows), dim3(256), 0, st, sims, ld, row_len, max_len, n_top, out_scores, out_ids);
  } else {
    return launch_topk_rows(sims, rows, max_len, ld, row_len, n_top, 1, out_scores, out_ids, st);  // n selection passes
  }
  FP_CHECK_LAUNCH("topn_rows");
  return FP_OK;
}

int launch_cosine_topk(const CosineArgs& a_in, int num_det, int num_obj, int max_det_per_obj, int max_templates, int n_top,
                       const int* det_num_templates, float* out_scores, int* out_ids, int tie_mode, hipStream_t st);

// scratch (floats) of the prefiltered retrieval, behind the FP_COSINE_SCRATCH_FLOATS region of the exact path:
//   approx [num_det, T] | exact keys [num_det, 8 parts, ceil(T / 8)] u64 | counts [num_det, 8] | any_flag
int launch_cosine_topk_prefiltered(const CosineArgs& a_in, int num_det, int num_obj, int max_det_per_obj, int max_templates, int n_top,
                                   const int* det_num_templates, float* out_scores, int* out_ids, int tie_mode, float* extra_scratch, hipStream_t st) {
  CosineArgs a = a_in;
  // Three dependent launches of ~10 us of latency each: the two-stage form pays for itself once the single-pass kernel has more than
  // ~250 MB of fp32 bank to stream -- many templates, or several 32-detection passes over them (measured: 10 000 templates x 32
  // detections 39 vs 40 us, 50 000 x 128 147 vs 319 us); below that the single-pass kernel stays the faster exact answer.
  const bool worth = a.force_prefilter || (size_t)max_templates * (size_t)cdiv(max_det_per_obj, 32) >= 30000;
  // The strict (torch) order replays tied rows from a whole row of EXACT scores, produced by cosine_fused_kernel<NQ, false> -- whose fp32
  // path holds at most 4 chunks per k-slice (W <= 2048; launch_cosine_topk sends wider banks to the generic kernel).  So with tie_mode 1
  // the two-stage form is taken for W <= 2048 only; W = 3072 / 4096 in the strict order run the single-pass exact path.
  const bool w_ok = a.W % 1024 == 0 && (tie_mode == 1 ? a.W <= 2048 : a.W <= 4096);
  const bool ok = worth && a.bank_bf16 && w_ok && max_templates <= COS_PARTS * COS_LIST_CAP && n_top + 1 <= COS_NMAX && max_det_per_obj >= 1;
  if (!ok) return launch_cosine_topk(a_in, num_det, num_obj, max_det_per_obj, max_templates, n_top, det_num_templates, out_scores, out_ids, tie_mode, st);
  const int tp = cdiv(max_templates, COS_PARTS);
  float* approx = extra_scratch;
  unsigned long long* exact_keys = reinterpret_cast<unsigned long long*>(approx + (size_t)num_det * max_templates + ((size_t)num_det * max_templates & 1));
  int* exact_cnt = reinterpret_cast<int*>(exact_keys + (size_t)num_det * COS_PARTS * tp);
  int* any_flag = exact_cnt + (size_t)num_det * COS_PARTS;
  a.k_slices = 8;
  const int nblk = cdiv(max_templates, 16), n_emit = n_top + 1;
  const int nq = max_det_per_obj <= 16 ? 1 : 2;
  const int chunks = cdiv(max_det_per_obj, nq * 16);
  const int per = fp_num_cus() / (num_obj * chunks);
  const int gx = per < 1 ? 1 : (per > nblk ? nblk : per);
  FP_REQUIRE(gx * n_emit <= 2048, "cosine_topk_prefiltered: too many candidate keys per detection");
  if (tie_mode == 1) {
    hipError_t e = hipMemsetAsync(any_flag, 0, sizeof(int), st);
    if (e != hipSuccess) { fp_set_error("cosine_topk_prefiltered: memset: %s", hipGetErrorString(e)); return FP_ERR_HIP; }
  }
  // (1) approximate pass over the fp16 bank
  CosineArgs b = a;
  b.sims = approx; b.n_top = n_emit; b.run_flag = nullptr;  // (b.cand: the exact path's key region, free until its fallback runs)
  const size_t lds = COS_RING_BYTES + (size_t)COS_RED_BUFS * 8 * nq * 4 * COS_RED_PITCH * 4;
  static FpDeviceOnce attr1, attr2;
  fp_allow_dynamic_lds(attr1, &cosine_fused_kernel<1, true>, COS_RING_BYTES + COS_RED_BUFS * 8 * 1 * 4 * COS_RED_PITCH * 4);
  fp_allow_dynamic_lds(attr2, &cosine_fused_kernel<2, true>, COS_RING_BYTES + COS_RED_BUFS * 8 * 2 * 4 * COS_RED_PITCH * 4);
  dim3 grid(gx, num_obj, chunks);
  if (nq == 1) hipLaunchKernelGGL((cosine_fused_kernel<1, true>), grid, dim3(512), lds, st, b);
  else hipLaunchKernelGGL((cosine_fused_kernel<2, true>), grid, dim3(512), lds, st, b);
  FP_CHECK_LAUNCH("cosine_fused<fp16>");
  // (2) candidates above (n+1)-th best - 2 eps, exact scores; (3) top n of the exact keys (+ tie flags in the strict order)
  hipLaunchKernelGGL(cosine_rescore_kernel, dim3(COS_PARTS, num_det), dim3(512), 0, st, a, b.cand, gx * n_emit, n_emit, approx, exact_keys, exact_cnt, tp);
  FP_CHECK_LAUNCH("cosine_rescore");
  hipLaunchKernelGGL(cosine_final_kernel, dim3(cdiv(num_det, 4)), dim3(256), 0, st, exact_keys, exact_cnt, num_det, n_top,
                     out_scores, out_ids, tie_mode == 1 ? a.need_replay : nullptr, any_flag, tp);
  FP_CHECK_LAUNCH("cosine_final");
  if (tie_mode == 0) return FP_OK;  // canonical order: (score, lowest id) is decided by the exact keys
  // strict (torch) order: rows with a tie among their best n + 1 scores need the whole row of exact scores for the replay
  CosineArgs f = a;
  f.cand = nullptr; f.run_flag = any_flag;
  static FpDeviceOnce attr3, attr4;
  fp_allow_dynamic_lds(attr3, &cosine_fused_kernel<1>, COS_RING_BYTES + COS_RED_BUFS * 8 * 1 * 4 * COS_RED_PITCH * 4);
  fp_allow_dynamic_lds(attr4, &cosine_fused_kernel<2>, COS_RING_BYTES + COS_RED_BUFS * 8 * 2 * 4 * COS_RED_PITCH * 4);
  if (nq == 1) hipLaunchKernelGGL(cosine_fused_kernel<1>, grid, dim3(512), lds, st, f);
  else hipLaunchKernelGGL(cosine_fused_kernel<2>, grid, dim3(512), lds, st, f);
  FP_CHECK_LAUNCH("cosine_fused(fallback)");
  return launch_topn_rows(a.sims, a.ld_sims, num_det, max_templates, det_num_templates, n_top, out_scores, out_ids, 1, st, a.need_replay);
}

int launch_cosine_topk(const CosineArgs& a_in, int num_det, int num_obj, int max_det_per_obj, int max_templates, int n_top,
                       const int* det_num_templates, float* out_scores, int* out_ids, int tie_mode, hipStream_t st) {
  CosineArgs a = a_in;
  FP_REQUIRE(a.W % 16 == 0, "cosine_topk: num_words %% 16 == 0 required on this path");
  a.k_slices = (a.W % 128 == 0) ? 8 : 1;  // canonical chain split, see include/foundpose_amd.h
  const int wslice = a.W / a.k_slices;
  const int nblk = cdiv(max_templates, 16);
  bool fused = a.k_slices == 8 && wslice % 64 == 0 && wslice <= 256;
  // canonical order: n_top candidates per workgroup finish the row.  torch order: n_top + 1, so that the merge can tell
  // whether the row has a tie at all (cand_merge_kernel); only rows that do are replayed from the scores
  const bool want_cand = tie_mode == 0 ? n_top <= COS_NMAX : n_top + 1 <= COS_NMAX;
  const int n_emit = tie_mode == 0 ? n_top : n_top + 1;
  if (fused) {
    // one persistent workgroup per CU, shared evenly by the (object, 32-detection chunk) pairs of the launch
    const int nq = max_det_per_obj <= 16 ? 1 : 2;
    const int chunks = cdiv(max_det_per_obj, nq * 16);
    const int per = fp_num_cus() / (num_obj * chunks);
    const int gx = per < 1 ? 1 : (per > nblk ? nblk : per);
    a.n_top = n_emit;
    if (!want_cand) a.cand = nullptr;
    const size_t lds = COS_RING_BYTES + (size_t)COS_RED_BUFS * 8 * nq * 4 * COS_RED_PITCH * 4;
    static FpDeviceOnce attr1, attr2;
    fp_allow_dynamic_lds(attr1, &cosine_fused_kernel<1>, COS_RING_BYTES + COS_RED_BUFS * 8 * 1 * 4 * COS_RED_PITCH * 4);
    fp_allow_dynamic_lds(attr2, &cosine_fused_kernel<2>, COS_RING_BYTES + COS_RED_BUFS * 8 * 2 * 4 * COS_RED_PITCH * 4);
    dim3 grid(gx, num_obj, chunks);
    if (nq == 1) hipLaunchKernelGGL(cosine_fused_kernel<1>, grid, dim3(512), lds, st, a);
    else hipLaunchKernelGGL(cosine_fused_kernel<2>, grid, dim3(512), lds, st, a);
    FP_CHECK_LAUNCH("cosine_fused");
    if (want_cand) {
#ifdef FP_EXPERIMENTS
      static const bool two_launches = getenv("FP_COSINE_MERGE_REPLAY") && atoi(getenv("FP_COSINE_MERGE_REPLAY")) == 0;  // A/B switch of measurement builds
#else
      constexpr bool two_launches = false;
#endif
      if (tie_mode == 1 && !two_launches && n_top <= 32) {  // merge + the replay of tied rows in one launch (torch order)
        hipLaunchKernelGGL(cand_merge_replay_kernel, dim3(num_det), dim3(256), 0, st, a.cand, gx * n_emit, num_det, n_top, out_scores, out_ids,
                           a.need_replay, a.sims, a.ld_sims, det_num_templates, max_templates);
        FP_CHECK_LAUNCH("cand_merge_replay");
        return FP_OK;
      }
      hipLaunchKernelGGL(cand_merge_kernel, dim3(num_det), dim3(256), 0, st, a.cand, gx * n_emit, num_det, n_top, out_scores, out_ids,
                         tie_mode == 1 ? a.need_replay : nullptr);
      FP_CHECK_LAUNCH("cand_merge");
      if (tie_mode == 0) return FP_OK;
      return launch_topn_rows(a.sims, a.ld_sims, num_det, max_templates, det_num_templates, n_top, out_scores, out_ids, tie_mode, st, a.need_replay);
    }
  } else {
    const int nq = max_det_per_obj <= 16 ? 1 : (max_det_per_obj <= 32 ? 2 : 4);
    const size_t lds = (size_t)nq * 16 * ((size_t)wslice * 4 + 16);
    FP_REQUIRE(lds <= 160 * 1024, "cosine_topk: query slice does not fit LDS (num_words %d)", a.W);
    static FpDeviceOnce attr;
    if (fp_first_on_device(attr)) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&cosine_generic_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&cosine_generic_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&cosine_generic_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    dim3 grid(cdiv(nblk, 4), num_obj, cdiv(max_det_per_obj, nq * 16));
    if (nq == 1) hipLaunchKernelGGL(cosine_generic_kernel<1>, grid, dim3(256), lds, st, a);
    else if (nq == 2) hipLaunchKernelGGL(cosine_generic_kernel<2>, grid, dim3(256), lds, st, a);
    else hipLaunchKernelGGL(cosine_generic_kernel<4>, grid, dim3(256), lds, st, a);
    FP_CHECK_LAUNCH("cosine_generic");
  }
  return launch_topn_rows(a.sims, a.ld_sims, num_det, max_templates, det_num_templates, n_top, out_scores, out_ids, tie_mode, st, nullptr);
}
