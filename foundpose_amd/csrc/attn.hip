// Fused multi-head self-attention for the DINOv2 blocks (head_dim 64, non-causal, N = 1+R+Np tokens).
//
// Replaces softmax(q k^T / sqrt(64)) v inside the backbone's Attention.forward (the reference reaches it
// through self.model(batch), /root/reference/utils/dinov2_utils.py:257; optional xformers path upstream).
//
// bf16 kernels (flash-style, one pass over the keys, fp32 online softmax).  attn_bf16_w64_kernel (default: 64 queries
// per wave, K/V tiles by LDS-DMA, see its header) and attn_bf16_kernel (32 queries per wave, register staging; kept as
// the cross-check, AttnArgs.variant = 1) produce bit-identical results.  Common to both:
//   * block = (128 queries, head, image), 4 waves x 32 queries; K tile and V tile (64 keys x 64 d each, rows of the
//     qkv buffer) staged through registers into LDS, next tile's loads in flight under the MFMAs, one barrier per tile
//   * S^T = K Q^T with v_mfma_f32_32x32x16_bf16 (operands swapped so a lane owns ONE query's scores:
//     row max / row sum need a single cross-lane exchange with lane^32)
//   * the K rows enter the score MFMA permuted -- A-row i holds key i with bits 2 and 3 swapped --, so output register r of lane (query, kh)
//     is key 16 (r >> 3) + 8 kh + (r & 7) of its 32-key half: eight consecutive keys per 16-key step, which IS the B operand layout of
//     O^T = V^T P^T.  P -> bf16 by v_cvt_pk_bf16_f32 where it sits; no cross-lane exchange (rounds 1-4 paired lanes by v_permlane32_swap:
//     16 swaps per wave and key tile, ~3 issue slots each)
//   * V stays row-major ([key][d], exactly as the qkv GEMM wrote it); the A operand V^T of the P*V MFMA is produced by
//     gfx950's hardware transpose read ds_read_b64_tr_b16 (two per fragment: lane (d = l&31, kh) receives keys
//     kh*8 + 4r + j of column d) from an LDS image cut into 16-column blocks.  No pre-transposed V^T copy in HBM
//     (92 MB per layer at the bench batch) and no transposing epilogue in the qkv GEMM.
//   * lazy rescale, decided per query: the exponent's reference moves only when a tile's maximum exceeds it by more than 8 (softmax is
//     shift-invariant; p <= 2^8), so the accumulators are rescaled in a handful of tiles instead of nearly all of them
// f16x3 kernel (attn_split_kernel): the same structure on split-fp16 operands, three fp16 MFMAs per product (its header).
// fp32 kernels (exact mode): attn_f32_mfma_kernel, flash attention on v_mfma_f32_32x32x2_f32 (its header), and attn_f32_kernel
// (one thread per query, one fma chain per score, exact expf; AttnArgs.variant = 1) as its cross-check.
#include <type_traits>
#include "common.hpp"
#include "kernels.hpp"

namespace {

FP_DEVICE int swz(int row) { return (row >> 1) & 7; }

FP_DEVICE bf16x8 read_frag(const char* lds, int row, int chunk) {
  return *reinterpret_cast<const bf16x8*>(lds + row * 128 + ((chunk ^ swz(row)) << 4));
}

// V image for the transpose reads: [4 blocks of 16 d][64 keys][16 d = 32 B]; block offsets 0 / 2176 / 4416 / 6592 so
// that (a) the two blocks a ds_read_b64_tr_b16 touches are 128 B apart mod 256 (lanes 0-31 cover all 64 banks once) and
// (b) the four blocks start in different bank quarters for the 16-B staging writes.
constexpr int VTR_DT = 4416, VTR_B = 2176, VTR_BYTES = 8704;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

__global__ __launch_bounds__(256, 4) void attn_bf16_kernel(AttnArgs a) {
  __shared__ __attribute__((aligned(16))) char KV[2][2][VTR_BYTES];  // [stage][K: 64 rows x 128 B swizzled | V: block image]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, kh = lane >> 5;
  const int krow = (l31 & 19) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);  // A-row i of the score MFMA holds key i with bits 2 and 3 swapped (header)
  // Workgroup -> (query tile, head, image).  Consecutive workgroup ids go round-robin over the 8 XCDs, each with its
  // own L2: when the (image, head) pairs divide by 8, all query tiles of a pair are given ids of ONE residue mod 8 so
  // that the pair's K and V (2 x N x 128 B) are fetched into one L2 instead of eight.
  int qt, head, img;
  {
    const int nqt = (a.n_tok + 127) / 128, pairs = a.heads * a.batch, i = blockIdx.x;
    int pair;
    if ((pairs & 7) == 0) {
      const int j = i >> 3;
      qt = j % nqt;
      pair = (j / nqt) * 8 + (i & 7);
    } else {
      qt = i % nqt;
      pair = i / nqt;
    }
    head = pair % a.heads;
    img = pair / a.heads;
  }
  const int N = a.n_tok, D = a.dim;
  const __bf16* qkv = reinterpret_cast<const __bf16*>(a.qkv) + (size_t)img * N * a.ld_qkv;

  // Q fragments straight from global (once per block): B operand, lane holds Q[query][8 d]
  const int q = qt * 128 + wave * 32 + l31;
  const int qc = q < N ? q : N - 1;
  bf16x8 qf[4];
#pragma unroll
  for (int ds = 0; ds < 4; ++ds)
    qf[ds] = *reinterpret_cast<const bf16x8*>(qkv + (size_t)qc * a.ld_qkv + head * 64 + (ds * 2 + kh) * 8);

  f32x16 oacc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[i][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const float c = 0.125f * 1.44269504088896340736f;  // head_dim^-0.5 * log2(e)

  // staging assignment: 2 K chunks + 2 V^T chunks (16 B each) per thread (rows srow0 and srow0 + 32)
  const int srow0 = tid >> 3, sch = tid & 7;
  const int soff0 = srow0 * 128 + ((sch ^ swz(srow0)) << 4);
  const int soff1 = (srow0 + 32) * 128 + ((sch ^ swz(srow0 + 32)) << 4);
  const __bf16* kbase = qkv + D + head * 64 + sch * 8;
  const __bf16* vbase = kbase + D;
  // V staging offsets: this thread's 8 d's (chunk sch) of key srow0 / srow0 + 32 inside the block image
  const int voff0 = (sch >> 2) * VTR_DT + ((sch >> 1) & 1) * VTR_B + srow0 * 32 + (sch & 1) * 16, voff1 = voff0 + 32 * 32;
  // transpose-read base: source chunk of this lane inside a [4 keys][16 d] block + its block (b = bit 4 of the lane) + key half
  const int vrd = ((lane >> 4) & 1) * VTR_B + (kh * 8 + ((lane & 15) >> 2)) * 32 + (lane & 3) * 8;
  uint4 kreg0, kreg1, vreg0, vreg1;
#define ATTN_LOAD_TILE(key0_)                                                                   \
  {                                                                                             \
    int k0_ = (key0_) + srow0, k1_ = (key0_) + srow0 + 32;                                      \
    k0_ = k0_ < N ? k0_ : N - 1;                                                                \
    k1_ = k1_ < N ? k1_ : N - 1;                                                                \
    kreg0 = *reinterpret_cast<const uint4*>(kbase + (size_t)k0_ * a.ld_qkv);                    \
    kreg1 = *reinterpret_cast<const uint4*>(kbase + (size_t)k1_ * a.ld_qkv);                    \
    vreg0 = *reinterpret_cast<const uint4*>(vbase + (size_t)k0_ * a.ld_qkv);                    \
    vreg1 = *reinterpret_cast<const uint4*>(vbase + (size_t)k1_ * a.ld_qkv);                    \
  }
#define ATTN_STORE_TILE(stage_)                                       \
  {                                                                   \
    *reinterpret_cast<uint4*>(KV[stage_][0] + soff0) = kreg0;         \
    *reinterpret_cast<uint4*>(KV[stage_][0] + soff1) = kreg1;         \
    *reinterpret_cast<uint4*>(KV[stage_][1] + voff0) = vreg0;         \
    *reinterpret_cast<uint4*>(KV[stage_][1] + voff1) = vreg1;         \
  }

  const int nkt = (N + 63) / 64;
  ATTN_LOAD_TILE(0);
  ATTN_STORE_TILE(0);
  __syncthreads();
  for (int kt = 0; kt < nkt; ++kt) {
    const int key0 = kt * 64;
    const char* Ks = KV[kt & 1][0];
    const char* Vs = KV[kt & 1][1];
    if (kt + 1 < nkt) ATTN_LOAD_TILE(key0 + 64);  // next tile: HBM/L2 -> registers, lands under this tile's MFMAs

    // ---- S^T = K Q^T : sacc[ks][r] = score(query = l31, key = key0 + ks*32 + 16*(r>>3) + 8*kh + (r&7)) -- A-row i holds key krow(i)
    f32x16 sacc[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[ks][r] = 0.f;
#pragma unroll
      for (int ds = 0; ds < 4; ++ds) {
        bf16x8 kf = read_frag(Ks, ks * 32 + krow, ds * 2 + kh);
        sacc[ks] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ds], sacc[ks], 0, 0, 0);
      }
    }
    if (key0 + 64 > N) {  // ragged last tile: mask the padded keys
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = key0 + ks * 32 + 16 * (r >> 3) + 8 * kh + (r & 7);
          if (key >= N) sacc[ks][r] = -INFINITY;
        }
    }
    // ---- online softmax (fp32). A query's 64 scores live in lanes l31 and l31+32.
    float mx = fmaxf(sacc[0][0], sacc[1][0]);
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(fmaxf(mx, sacc[0][r]), sacc[1][r]);  // v_max3_f32
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    // lazy rescale, per query: the exponent's reference m_run moves only when the tile maximum exceeds it by more than 8 in the
    // exponent (see attn_bf16_w64_kernel; the same rule, so the two kernels stay bit-identical)
#ifdef FP_ATTN_EAGER_RESCALE
    const bool moves = mx > m_run;
#else
    const bool moves = mx - m_run > 8.f / (0.125f * 1.44269504088896340736f);
#endif
    const bool grow = __any(moves);
    float alpha = 1.f;
    if (grow) {
      const float m_new = moves ? mx : m_run;
      alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
      m_run = m_new;
    }
    // exp2(s*c - m*c): scalar fp32 VALU on purpose -- v_pk_fma/v_pk_add variants measured 25 % SLOWER here
    // (packed fp32 issues badly beside MFMAs, guide "price of one filler beside MFMAs")
    float psum = 0.f;
    const float mc = m_run * c;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = __builtin_amdgcn_exp2f(fmaf(sacc[ks][r], c, -mc));
        sacc[ks][r] = p;
        psum += p;
      }
    if (grow) {
      l_run *= alpha;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[i][r] *= alpha;
    }
    l_run += psum;

    // ---- O^T += V^T P^T over 4 steps of 16 keys
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const int r0 = 8 * kk;
        // registers r0 .. r0 + 7 are eight consecutive keys (the K rows' order, header): the B operand as it stands
        uint4 pw = make_uint4(pack_bf16x2(sacc[ks][r0 + 0], sacc[ks][r0 + 1]), pack_bf16x2(sacc[ks][r0 + 2], sacc[ks][r0 + 3]),
                              pack_bf16x2(sacc[ks][r0 + 4], sacc[ks][r0 + 5]), pack_bf16x2(sacc[ks][r0 + 6], sacc[ks][r0 + 7]));
        bf16x8 pf = __builtin_bit_cast(bf16x8, pw);
        const int kstep = ks * 2 + kk;  // keys kstep*16 .. +15 of the tile
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const char* vp = Vs + vrd + dt * VTR_DT + kstep * 512;  // 16 keys x 32 B per k-step; + 128 B = 4 keys on
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(vp));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(vp + 128));
          const bf16x8 vf = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
          oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, oacc[dt], 0, 0, 0);
        }
      }
    if (kt + 1 < nkt) ATTN_STORE_TILE((kt + 1) & 1);  // the other stage was last read one iteration ago
    __syncthreads();
  }

  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.f / l_tot;
  if (q < N) {
    __bf16* o = reinterpret_cast<__bf16*>(a.out) + ((size_t)img * N + q) * a.ld_out + head * 64;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = dt * 32 + 8 * g + 4 * kh;
        uint2 pk = make_uint2(pack_bf16x2(oacc[dt][4 * g + 0] * inv, oacc[dt][4 * g + 1] * inv),
                              pack_bf16x2(oacc[dt][4 * g + 2] * inv, oacc[dt][4 * g + 3] * inv));
        *reinterpret_cast<uint2*>(o + d) = pk;
      }
  }
}

// ---------------------------------------------------------------- bf16, 64 queries per wave
// Same arithmetic as attn_bf16_kernel (identical MFMA operands and order per query => bit-identical output), different
// work split: block = 256 queries = 4 waves x TWO 32-query blocks.  A K fragment (ds_read_b128) and a V^T fragment
// (two transpose reads) feed two MFMAs instead of one, and a staged K/V tile serves 256 queries instead of 128, which
// halves the LDS read, LDS write and L1 fill traffic per flop -- the old split kept the LDS pipe ~80 % busy.  K and V
// tiles arrive by LDS-DMA (buffer_load_dwordx4 ... lds, whole 128-B rows, no staging registers, no ds_write):
//   K image  [64 keys][128 B], 16-B chunk p of row r holds chunk p ^ ((r >> 1) & 7)            (b128 fragment reads)
//   V image  [64 keys][128 B], 64-B half  g of row r holds half  g ^ ((r >> 1) & 1)            (transpose reads: the
//            four keys x 64 B a half-wave touches then fall into four different bank quarters)
// Rows past the last token read as zeros (buffer range check) and are masked exactly like before.
typedef __attribute__((address_space(3))) void lds_void_t;

// QB = 32-query blocks per wave: 2 (4 waves per block, 208 VGPRs, 2 waves/SIMD -- the default) or 1 (8 waves per block of the
// same 256 queries, <= 128 VGPRs, 4 waves/SIMD: more waves to overlap, twice the LDS fragment traffic per flop).
// NW = waves per block (default 512 / (64 QB): a 256-query block); QB = 2 with NW = 8 is a 512-query block: a staged K/V tile then serves
// twice the queries (variant 3, measured in DESIGN section 5).  Each wave stages GW = 8 / NW of the tile's eight 8-key row groups.
// PF (variant 4, measurement): the NEXT tile's eight K fragments are read into registers under the current tile's P V MFMAs (its DMA was
// issued at the top of this tile and has landed by then: the compiler drains vmcnt before the first transpose read anyway; one extra
// barrier publishes the other waves' pieces), so the next tile's score MFMAs start without waiting for LDS.  Same arithmetic.
// H16 ("f16" mode): q | k | v, P and the output are IEEE fp16 instead of bf16 (v_mfma_f32_32x32x16_f16; p <= 2^8 and a convex combination of v rows
// both fit the format); everything else -- layouts, schedule, fp32 online softmax -- is shared.
template <int QB, int NW = 8 / QB, bool PF = false, bool H16 = false>
__global__ __launch_bounds__(NW * 64, 2) void attn_bf16_w64_kernel(AttnArgs a) {
  constexpr int QBLK = NW * 32 * QB, GW = 8 / NW;
  __shared__ __attribute__((aligned(16))) char KV[2][2][8192];  // [stage][K | V]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: DMA offsets
  const int l31 = lane & 31, kh = lane >> 5;
  const int krow = (l31 & 19) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);  // as in attn_bf16_kernel
  int qt, head, img;
  {
    const int nqt = ((a.sel_off ? a.max_sel : a.n_tok) + QBLK - 1) / QBLK, pairs = a.heads * a.batch, i = blockIdx.x;
    int pair;
    if ((pairs & 7) == 0) {  // all query tiles of an (image, head) pair on one XCD (one L2), see attn_bf16_kernel
      const int j = i >> 3;
      if (a.tail_last && nqt > 1) {  // ... the pair's last tile (short: half a block's work or less) at the END of the XCD's sequence
        const int nfull = (pairs >> 3) * (nqt - 1);
        const bool tail = j >= nfull;
        qt = tail ? nqt - 1 : j % (nqt - 1);
        pair = (tail ? j - nfull : j / (nqt - 1)) * 8 + (i & 7);
      } else {
        qt = j % nqt;
        pair = (j / nqt) * 8 + (i & 7);
      }
    } else {
      qt = i % nqt;
      pair = i / nqt;
    }
    head = pair % a.heads;
    img = pair / a.heads;
  }
  const int N = a.n_tok, D = a.dim;
  const __bf16* qkv = reinterpret_cast<const __bf16*>(a.qkv) + (size_t)img * N * a.ld_qkv;
  // queries: every token, or the image's selected tokens (the keys are always all N tokens)
  const int sel_base = a.sel_off ? a.sel_off[img] : 0;
  const int NQ = a.sel_off ? a.sel_off[img + 1] - sel_base : N;
  if (qt * QBLK >= NQ) return;  // selected mode: the grid is sized for the image with the most queries
  // The last query tile of an (image, head) pair is usually short (1374 tokens = 5 x 256 + 94).  With two 32-query blocks
  // per wave only 2 of its 4 waves would have queries, each doing a full tile's work: the block would cost as much as a
  // full one for 37 % of the queries.  When <= 128 queries remain every wave takes ONE 32-query block instead (the QC = 1
  // instantiation of the tile loop): the same arithmetic per query, half the work per wave, the tail block ends in about
  // half the time.  Block-uniform.  (Running the launch's LAST blocks as such half blocks, to fill the drain of the final round with
  // twice as many blocks of half the lifetime, measured slower -- 345 / 353 / 363 us for the last 32 / 64 / 128 workgroup slots per
  // XCD against 344 without: the half blocks stage every K / V tile for half the queries.)
  const bool short_tail = QB == 2 && qt == (NQ + QBLK - 1) / QBLK - 1 && NQ - qt * QBLK <= QBLK / 2;
  const int nqb = short_tail ? 1 : QB;
  const int q0 = qt * QBLK + wave * (32 * nqb);
  const bool active = q0 < NQ;  // wave-uniform; an inactive wave only stages tiles and keeps the barriers

  // ---- staging: wave w issues row groups 2w, 2w+1 (8 keys x 128 B each) of K and of V
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)qkv, 0, (unsigned)((size_t)N * a.ld_qkv * 2), 0x00020000);
  const int sr = lane >> 3, sp = lane & 7;
  const unsigned rowoff = (unsigned)(sr * a.ld_qkv) * 2u;
  const unsigned voff_k0 = rowoff + ((sp ^ (sr >> 1)) << 4);        // even row group: swizzle (r >> 1) & 7 = sr >> 1
  const unsigned voff_k1 = rowoff + ((sp ^ ((sr >> 1) + 4)) << 4);  // odd row group: ... + 4
  const unsigned voff_v = rowoff + ((sp ^ (((sr >> 1) & 1) << 2)) << 4);
  const unsigned tile_stride = (unsigned)(64 * a.ld_qkv) * 2u, grp_stride = (unsigned)(8 * a.ld_qkv) * 2u;
  const unsigned soff_k = (unsigned)(D + head * 64) * 2u + (unsigned)GW * wave * grp_stride, soff_v = soff_k + (unsigned)D * 2u;
  const unsigned voff_kw = (wave & 1) ? voff_k1 : voff_k0;  // GW == 1: wave w stages row group w only
  auto stage_tile = [&](int kt, int stage) {
    const unsigned t = kt * tile_stride;
    char* kd = KV[stage][0] + wave * (1024 * GW);
    char* vd = KV[stage][1] + wave * (1024 * GW);
    if constexpr (GW == 2) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_t*)kd, 16, voff_k0, soff_k + t, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_t*)(kd + 1024), 16, voff_k1, soff_k + t + grp_stride, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_t*)vd, 16, voff_v, soff_v + t, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_t*)(vd + 1024), 16, voff_v, soff_v + t + grp_stride, 0, 0);
    } else {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_t*)kd, 16, voff_kw, soff_k + t, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_t*)vd, 16, voff_v, soff_v + t, 0, 0);
    }
  };
  const int nkt = (N + 63) / 64;
  stage_tile(0, 0);

  const float c = 0.125f * 1.44269504088896340736f;  // head_dim^-0.5 * log2(e)
  constexpr float LAZY_TH = 8.f / (0.125f * 1.44269504088896340736f);  // 8 in the exponent, in score units
  // Q fragments straight from global (once per block): B operand, lane holds Q[query][8 d]
  bf16x8 qf[QB][4];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    const int q = q0 + (qb < nqb ? qb : 0) * 32 + l31;
    int qc = q < NQ ? q : NQ - 1;
    if (a.sel_rows) qc = a.sel_rows[sel_base + qc] - img * N;  // the selected query's token
#pragma unroll
    for (int ds = 0; ds < 4; ++ds)
      qf[qb][ds] = *reinterpret_cast<const bf16x8*>(qkv + (size_t)qc * a.ld_qkv + head * 64 + (ds * 2 + kh) * 8);
  }
  f32x16 oacc[QB][2];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[qb][i][r] = 0.f;
  float m_run[QB], l_run[QB];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) { m_run[qb] = -INFINITY; l_run[qb] = 0.f; }
  // transpose-read base inside the V image: key = kh*8 + kq, 16-d block b, 8-B piece; the 64-B half flips with kq >> 1
  const int kq = (lane & 15) >> 2, vb = (lane >> 4) & 1;
  const int vrd0 = (kh * 8 + kq) * 128 + (((kq >> 1) & 1) << 6) + vb * 32 + (lane & 3) * 8;

  // s_waitcnt as a BUILTIN (vmcnt(0), encoding 0x0f70): the compiler's wait-count pass sees it and knows the Q fragment
  // loads have landed -- behind an opaque asm it re-waited for them inside the loop, with counts that also drained
  // the LDS-DMA of the iteration (one full load latency exposed per tile)
  __builtin_amdgcn_s_waitcnt(0x0f70);
  __syncthreads();
  bf16x8 kfn[PF ? 4 : 1][2];   // PF: the K fragments of the tile about to be scored
  if constexpr (PF) {
#pragma unroll
    for (int ds = 0; ds < 4; ++ds)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) kfn[ds][ks] = read_frag(KV[0][0], ks * 32 + krow, ds * 2 + kh);
  }
  // one key tile; RAGGED (the last tile when N % 64 != 0) is a separate instantiation so that the full tiles carry no
  // masking code at all (inlined into one loop the compiler if-converts the mask into 120 selects per tile)
  auto tile = [&](int kt, auto ragged, auto qblocks, auto halfkeys) {
    constexpr bool RAGGED = decltype(ragged)::value;
    constexpr int QC = decltype(qblocks)::value;  // 32-query blocks this wave computes (QB, or 1 in a short tail tile)
    // HALF: the ragged last tile holds <= 32 live keys (1374 tokens = 21 x 64 + 30): its second key half is all padding -- scores -inf,
    // probabilities 0, V rows read as zeros -- and is skipped: the same sums without their zero addends
    constexpr int KS = decltype(halfkeys)::value ? 1 : 2;
    static_assert(RAGGED || KS == 2, "only the ragged tile can be half empty");
    const int key0 = kt * 64;
    const char* Ks = KV[kt & 1][0];
    const char* Vs = KV[kt & 1][1];
#ifndef FP_ATTN_NO_DMA  // (measurement builds, tools/attn_ablate.sh: the tile loop without its K/V stream -- every tile re-reads tile 0's image)
    // the other stage was last read one iteration ago.  (Issued from inside the softmax instead -- between the two query blocks, a
    // VALU-only stretch -- the kernel measured 1.5 % SLOWER: 362 vs 356 us; issued after the tile's first two K fragment reads: 344 vs 338.)
    if (!RAGGED && kt + 1 < nkt) stage_tile(kt + 1, (kt + 1) & 1);
#endif
    bf16x8 pf[QC][4];
    if (active) {
      if constexpr (!PF) __builtin_amdgcn_iglp_opt(1);  // the compiler's MFMA / LDS interleaving strategy 1 for the tile body: +0.5 % same-box (0, 2, 3: -1...-2 %); (its solver does not terminate on the PF body)
      // ---- S^T = K Q^T for both query blocks: sacc[qb][ks][r] = score(query l31 of block qb, key key0 + ks*32 + 16*(r>>3) + 8*kh + (r&7))
      f32x16 sacc[QC][2];
      // four independent accumulation chains (2 key halves x 2 query blocks) interleaved over the four 16-d steps (two chains, key
      // half outermost, measured 1 % slower: a dependent MFMA waits for its predecessor's last pass)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int qb = 0; qb < QC; ++qb)
#pragma unroll
          for (int r = 0; r < 16; ++r) sacc[qb][ks][r] = 0.f;
#pragma unroll
      for (int ds = 0; ds < 4; ++ds)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
#ifdef FP_ATTN_NO_LDS  // (measurement builds: fragments from registers instead of LDS)
          bf16x8 kf = qf[0][ds];
          kf[0] = (__bf16)(float)((kt + ks) & 3);
#else
          const bf16x8 kf = PF ? kfn[PF ? ds : 0][ks] : read_frag(Ks, ks * 32 + krow, ds * 2 + kh);  // one K fragment, two MFMAs
#endif
#pragma unroll
          for (int qb = 0; qb < QC; ++qb)
            sacc[qb][ks] = mfma_h<H16>(kf, qf[qb][ds], sacc[qb][ks]);
        }
#if !defined(FP_ATTN_NO_LDS) && defined(FP_ATTN_TR_ASM)
      // (-DFP_ATTN_TR_ASM, measured and NOT the default.)  The tile's eight V^T fragments issued HERE (they land under the softmax)
      // and as inline asm: through the builtin the compiler cannot tell a transpose read from an access to the stage the LDS-DMA
      // is filling and puts s_waitcnt vmcnt(0) in front of the first one -- the prefetch of tile t+1 has to land in the middle of
      // tile t.  (LDS returns in order, so the compiler's own lgkmcnt waits for its K reads stay conservative with these in the
      // queue.)  Same speed within the noise of one box (347.6 / 347.0 us vs 341.5 / 350.0), 235 instead of 208 VGPRs.
      s16x4 vlo[4][2], vhi[4][2];
      {
        const unsigned vs0 = (unsigned)(size_t)(const __attribute__((address_space(3))) char*)(Vs + vrd0);
        const unsigned vs1 = (unsigned)(size_t)(const __attribute__((address_space(3))) char*)(Vs + (vrd0 ^ 64));
#pragma unroll
        for (int kstep = 0; kstep < 4; ++kstep) {
          asm volatile("ds_read_b64_tr_b16 %0, %4 offset:%6\n\tds_read_b64_tr_b16 %1, %4 offset:%7\n\t"
                       "ds_read_b64_tr_b16 %2, %5 offset:%6\n\tds_read_b64_tr_b16 %3, %5 offset:%7"
                       : "=&v"(vlo[kstep][0]), "=&v"(vhi[kstep][0]), "=&v"(vlo[kstep][1]), "=&v"(vhi[kstep][1])
                       : "v"(vs0), "v"(vs1), "n"(kstep * 2048), "n"(kstep * 2048 + 512));
        }
      }
#endif
#pragma unroll
      for (int qb = 0; qb < QC; ++qb) {
        if constexpr (RAGGED) {  // mask the padded keys (one lane-dependent limit, constant offsets)
          const int lim = N - key0 - 8 * kh;
#pragma unroll
          for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int r = 0; r < 16; ++r)
              if (ks * 32 + 16 * (r >> 3) + (r & 7) >= lim) sacc[qb][ks][r] = -INFINITY;
        }
        // ---- online softmax (fp32). A query's 64 scores live in lanes l31 and l31+32.
        float mx = KS == 2 ? fmaxf(sacc[qb][0][0], sacc[qb][KS - 1][0]) : sacc[qb][0][0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = KS == 2 ? fmaxf(fmaxf(mx, sacc[qb][0][r]), sacc[qb][KS - 1][r]) : fmaxf(mx, sacc[qb][0][r]);  // v_max3_f32
        {  // the query's other 32 scores live in lane ^ 32: one v_permlane32_swap (VALU) instead of a ds_bpermute round trip through LDS
          const unsigned mu = __builtin_bit_cast(unsigned, mx);
          const auto sw = __builtin_amdgcn_permlane32_swap(mu, mu, false, false);  // lanes < 32: {own, partner's}; >= 32: {partner's, own}
          const unsigned m0 = sw[0], m1 = sw[1];  // (through temporaries: __builtin_bit_cast applied to sw[1] directly reads element 0 with this hipcc)
          mx = fmaxf(__builtin_bit_cast(float, m0), __builtin_bit_cast(float, m1));
        }
        // Lazy rescale: m_run is the REFERENCE of the exponent, not necessarily the running maximum.  It moves (and O, l are rescaled)
        // only when some lane's tile maximum exceeds it by more than LAZY_TH, i.e. when a probability would exceed 2^8; softmax is
        // shift-invariant, so the result is the same function of the scores, with p <= 256 instead of <= 1 (fp32 sums, bf16 P:
        // the same relative precision).  With the exact maximum some lane of 64 sees a new one in most tiles (random scores:
        // 1 - (1 - 1/t)^32) and the wave pays the rescale of its 64 accumulator registers nearly every tile.
        // The decision is per QUERY (a lane whose query does not move multiplies by exactly 1): a query's result does not depend
        // on which other queries share its wave -- selected-token runs stay bit-identical to the full forward.
#ifdef FP_ATTN_EAGER_RESCALE
        const bool moves = mx > m_run[qb];
#else
        const bool moves = mx - m_run[qb] > LAZY_TH;
#endif
        const bool grow = __any(moves);
        float alpha = 1.f;
        if (grow) {
          const float m_new = moves ? mx : m_run[qb];
          alpha = __builtin_amdgcn_exp2f((m_run[qb] - m_new) * c);
          m_run[qb] = m_new;
        }
        float psum = 0.f;
        const float mc = m_run[qb] * c;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float p = __builtin_amdgcn_exp2f(fmaf(sacc[qb][ks][r], c, -mc));
            sacc[qb][ks][r] = p;
            psum += p;
          }
        if (grow) {
          l_run[qb] *= alpha;
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[qb][i][r] *= alpha;
        }
        l_run[qb] += psum;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
          for (int kk = 0; kk < 2; ++kk) {
            const int r0 = 8 * kk;
            // eight consecutive keys per lane and 16-key step (the K rows' order): P is packed where it is, no cross-lane exchange
            pf[qb][ks * 2 + kk] = __builtin_bit_cast(bf16x8, make_uint4(pack_h2<H16>(sacc[qb][ks][r0 + 0], sacc[qb][ks][r0 + 1]), pack_h2<H16>(sacc[qb][ks][r0 + 2], sacc[qb][ks][r0 + 3]),
                                                                         pack_h2<H16>(sacc[qb][ks][r0 + 4], sacc[qb][ks][r0 + 5]), pack_h2<H16>(sacc[qb][ks][r0 + 6], sacc[qb][ks][r0 + 7])));
          }
      }
      // ---- O^T += V^T P^T over 4 steps of 16 keys; one V^T fragment serves both query blocks
    }
    if constexpr (PF && !RAGGED) {
      if (kt + 1 < nkt) {     // (block-uniform, inactive waves included) the next tile's DMA has landed in every wave: publish it
        __builtin_amdgcn_s_waitcnt(0x0f70);
        __syncthreads();
      }
    }
    if (active) {
      if constexpr (PF && !RAGGED) {
        if (kt + 1 < nkt) {   // ... and read its K fragments under the P V MFMAs below
          const char* Kn = KV[(kt + 1) & 1][0];
#pragma unroll
          for (int ds = 0; ds < 4; ++ds)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) kfn[ds][ks] = read_frag(Kn, ks * 32 + krow, ds * 2 + kh);
        }
      }
#if !defined(FP_ATTN_NO_LDS) && defined(FP_ATTN_TR_ASM)
      asm volatile("s_waitcnt lgkmcnt(0)"  // the asm reads above (the compiler does not count them)
                   : "+v"(vlo[0][0]), "+v"(vhi[0][0]), "+v"(vlo[0][1]), "+v"(vhi[0][1]), "+v"(vlo[1][0]), "+v"(vhi[1][0]), "+v"(vlo[1][1]), "+v"(vhi[1][1]),
                     "+v"(vlo[2][0]), "+v"(vhi[2][0]), "+v"(vlo[2][1]), "+v"(vhi[2][1]), "+v"(vlo[3][0]), "+v"(vhi[3][0]), "+v"(vlo[3][1]), "+v"(vhi[3][1]));
#endif
#pragma unroll
      for (int kstep = 0; kstep < 2 * KS; ++kstep)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const char* vp = Vs + (vrd0 ^ (dt << 6)) + kstep * 2048;  // + 512 B = 4 keys on
#ifdef FP_ATTN_NO_LDS
          bf16x8 vf = qf[0][kstep];
          vf[0] = (__bf16)(float)((kt + dt) & 3);
          (void)vp;
#elif !defined(FP_ATTN_TR_ASM)
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(vp));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(vp + 512));
          const bf16x8 vf = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
#else
          (void)vp;
          const bf16x8 vf = __builtin_bit_cast(bf16x8, __builtin_shufflevector(vlo[kstep][dt], vhi[kstep][dt], 0, 1, 2, 3, 4, 5, 6, 7));
#endif
#pragma unroll
          for (int qb = 0; qb < QC; ++qb)
            oacc[qb][dt] = mfma_h<H16>(vf, pf[qb][kstep], oacc[qb][dt]);
        }
    }
#ifndef FP_ATTN_NO_BARRIER  // (measurement builds: no per-tile wait + barrier; only meaningful together with FP_ATTN_NO_DMA)
    if constexpr (!RAGGED) {
      __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0): the compiler does not wait for LDS-DMA before a barrier
      __syncthreads();
    }
#endif
  };
  const int nfull = N / 64;
  auto run = [&](auto qblocks) {
    for (int kt = 0; kt < nfull; ++kt) tile(kt, std::false_type{}, qblocks, std::false_type{});
    if (nfull < nkt) {
#ifndef FP_ATTN_NO_HALF  // (measurement build: the ragged tile always at full width)
      if (!PF && N - nfull * 64 <= 32) tile(nfull, std::true_type{}, qblocks, std::true_type{});
      else
#endif
      tile(nfull, std::true_type{}, qblocks, std::false_type{});
    }
  };
  if constexpr (QB == 2) {
    if (short_tail) run(std::integral_constant<int, 1>{});
    else run(std::integral_constant<int, 2>{});
  } else {
    run(std::integral_constant<int, QB>{});
  }

  if (active) {
    float sat_amax = 0.f;  // fp8 output: largest |scale * o| quantised (saturation report)
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      if (qb >= nqb) break;
      const int q = q0 + qb * 32 + l31;
      const float l_tot = l_run[qb] + __shfl_xor(l_run[qb], 32, 64);
      const float inv = 1.f / l_tot;
      const size_t orow = a.sel_off ? (size_t)(sel_base + q) : (size_t)img * N + q;  // compact rows in selected mode
      if (q < NQ && a.out_fp8_scale > 0.f) {  // fp8 mode: the proj GEMM's input, quantised here (ld_out in bytes)
        unsigned char* o = reinterpret_cast<unsigned char*>(a.out) + orow * a.ld_out + head * 64;
        const float sc = inv * a.out_fp8_scale;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
          for (int g = 0; g < 4; ++g)
            *reinterpret_cast<unsigned*>(o + dt * 32 + 8 * g + 4 * kh) =
                pack_fp8x4(oacc[qb][dt][4 * g + 0] * sc, oacc[qb][dt][4 * g + 1] * sc, oacc[qb][dt][4 * g + 2] * sc, oacc[qb][dt][4 * g + 3] * sc, sat_amax);
      } else if (a.out_fp8_scale <= 0.f) {
        // A lane holds 4 consecutive d of its query per (dt, g) and lane ^ 32 the next 4: one v_permlane32_swap per register pairs
        // them into 8 consecutive d = one 16-B store per lane (8 dwordx4 stores per query block instead of 16 dwordx2: the store
        // tail of an attention block is issue-bound, guide "attention epilogue store tail").  The swap runs with all lanes on.
        __bf16* o = reinterpret_cast<__bf16*>(a.out) + orow * a.ld_out + head * 64;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int g0 = 2 * j, g1 = 2 * j + 1;
            const unsigned x0 = pack_h2<H16>(oacc[qb][dt][4 * g0 + 0] * inv, oacc[qb][dt][4 * g0 + 1] * inv);
            const unsigned x1 = pack_h2<H16>(oacc[qb][dt][4 * g0 + 2] * inv, oacc[qb][dt][4 * g0 + 3] * inv);
            const unsigned y0 = pack_h2<H16>(oacc[qb][dt][4 * g1 + 0] * inv, oacc[qb][dt][4 * g1 + 1] * inv);
            const unsigned y1 = pack_h2<H16>(oacc[qb][dt][4 * g1 + 2] * inv, oacc[qb][dt][4 * g1 + 3] * inv);
            const auto s0 = __builtin_amdgcn_permlane32_swap(x0, y0, false, false);  // lanes < 32: {own x, partner's x}; >= 32: {partner's y, own y}
            const auto s1 = __builtin_amdgcn_permlane32_swap(x1, y1, false, false);
            if (q < NQ) *reinterpret_cast<uint4*>(o + dt * 32 + 16 * j + 8 * kh) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
          }
      }
    }
    if (a.out_fp8_scale > 0.f) report_saturation(a.sat, 1, sat_amax, FP_E4M3_MAX);
  }
}

// ---------------------------------------------------------------- split-fp16 (f16x3 mode): near-exact attention on the fp16 MFMA
// Same structure as attn_bf16_w64_kernel<1> (8 waves x one 32-query block, K/V tiles by LDS-DMA, S^T = K Q^T so a lane owns one
// query's scores, V^T by transpose reads), but every operand is a split-fp16 pair (common.hpp) and every product three MFMAs
// (hi*hi + hi*lo + lo*hi, fp32 accumulate): q, k, v arrive as the split rows the qkv GEMM wrote (scale SQ each), P is split in
// registers (scale 2^14; its row sum is taken from the unsplit fp32 values), the output leaves as a split row (scale out_scale).
//   qkv row: [q | k | v], each 2D halves; head h = halves [128 h, 128 h + 128) of its part = two 32-d groups of [hi 32 | lo 32]
//   K image [64 keys][256 B]: 16-B chunk p of row r holds chunk p ^ (r & 15)        (b128 fragment reads: 16 lanes, 16 rows, 16 slots)
//   V image [64 keys][256 B]: 64-B unit  u of row r holds unit  u ^ (r & 3)         (transpose reads: a half-wave's four keys x 64 B
//                                                                                    fall into the four bank quarters)
// 1 in the exponent for UNSCALED scores, in score units.  The scores carry in_scale^2 (c = 0.125 log2 e / in_scale^2), so in exponent units the reference
// trails the running maximum by at most 1 / in_scale^2 (1/256 at the pipeline's in_scale = 16: the rescale fires on most increases of the maximum -- kept,
// because the threshold is part of the mode's recorded arithmetic).  p <= 2 -- what the unclamped packing of p' = 2^14 p relies on -- holds for in_scale >= 1,
// which attn_launch / fp_attention_split require.
constexpr float SPLIT_LAZY_TH = 1.f / (0.125f * 1.44269504088896340736f);

// Per-wave state and the three per-tile phases both split-fp16 kernels run (the schedules differ, the arithmetic does not: bit-identical outputs).
//   scores():  S^T = K Q^T for 64 keys.  A-row i of a 32-key half holds key i with bits 2 and 3 swapped, so that output register r of lane
//              (query l31, half kh) is key 16 (r >> 3) + 8 kh + (r & 7) of its half: eight consecutive keys per 16-key step, which is the B
//              operand layout of the P V MFMA -- P is packed where it is, no cross-lane exchange.
//   softmax(): online softmax, lazily rescaled per query (the reference trails the maximum by at most 1 in the exponent: p <= 2); the scale 2^14
//              of the split P rides in the exponent's offset (p' = 2^14 p <= 2^15 fits fp16), the row sum is taken from p' in four chains;
//              p' -> (hi, lo) = (f16(p'), f16(p' - hi)) by one v_cvt_pk_f16_f32 per pair + one v_fma_mix{lo,hi}_f16 per value.
//   pv():      O^T += V^T P^T, V^T by transpose reads; per accumulator lo.hi, hi.lo, hi.hi per 16-key step, steps ascending.
constexpr float SPLIT_P_LOG2_SCALE = 14.f;
struct SplitAttnWave {
  f16x8 qh[4], ql[4];
  f32x16 sacc[2], oacc[2];
  f16x8 ph[4], pl[4];
  float m_run, l_run, c;
  int l31, kh, krow, kq, vrd0;

  FP_DEVICE void init(int lane, float in_scale) {
    l31 = lane & 31, kh = lane >> 5;
    krow = (l31 & 19) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);
    kq = (lane & 15) >> 2;
    vrd0 = (kh * 8 + kq) * 256 + ((lane >> 4) & 1) * 32 + (lane & 3) * 8;
    c = 0.125f * 1.44269504088896340736f / (in_scale * in_scale);  // exp2 argument = score * head_dim^-0.5 * log2(e); the scores carry in_scale^2
    m_run = -INFINITY, l_run = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[i][r] = 0.f, sacc[i][r] = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) ph[i] = f16x8{0, 0, 0, 0, 0, 0, 0, 0}, pl[i] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
  }
  // Q fragments straight from global (once per block): B operand, lane holds Q[query][8 d] as (hi, lo)
  FP_DEVICE void load_q(const _Float16* qp) {
#pragma unroll
    for (int ds = 0; ds < 4; ++ds) {
      const int off = (ds >> 1) * 64 + (ds & 1) * 16 + kh * 8;
      qh[ds] = *reinterpret_cast<const f16x8*>(qp + off);
      ql[ds] = *reinterpret_cast<const f16x8*>(qp + off + 32);
    }
  }
  // keys_left: N - (first key of the tile); RAGGED: the tile has fewer than 64 live keys, the rest are masked (their K rows are the DMA's
  // out-of-range zeros).  A compile-time flag: as a run-time test the compiler turns the masking into 32 selects in EVERY tile.
  template <bool RAGGED>
  FP_DEVICE void scores(const char* Ks, int keys_left) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[ks][r] = 0.f;
#pragma unroll
    for (int ds = 0; ds < 4; ++ds) {
      const int ch = (ds >> 1) * 8 + (ds & 1) * 2 + kh;  // hi chunk of this lane's 8 d; lo chunk 4 further
      f16x8 kfh[2], kfl[2];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int row = ks * 32 + krow;
        const char* kr = Ks + row * 256;
#ifdef SPP_NO_LDS   // measurement build: fragments from registers (loop-variant through keys_left, so nothing is hoisted or merged)
        kfh[ks] = qh[(ds + ks) & 3]; kfl[ks] = ql[(ds + ks + 1) & 3];
        kfh[ks][0] = (_Float16)(float)(keys_left & 7); kfl[ks][1] = (_Float16)(float)((keys_left >> 3) & 7);
        (void)kr;
#else
        kfh[ks] = *reinterpret_cast<const f16x8*>(kr + ((ch ^ (row & 15)) << 4));
        kfl[ks] = *reinterpret_cast<const f16x8*>(kr + (((ch + 4) ^ (row & 15)) << 4));
#endif
      }
      // the two key halves' chains interleaved (per accumulator the order is lo.hi, hi.lo, hi.hi over ds ascending)
      sacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kfl[0], qh[ds], sacc[0], 0, 0, 0);
      sacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kfl[1], qh[ds], sacc[1], 0, 0, 0);
      sacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kfh[0], ql[ds], sacc[0], 0, 0, 0);
      sacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kfh[1], ql[ds], sacc[1], 0, 0, 0);
      sacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kfh[0], qh[ds], sacc[0], 0, 0, 0);
      sacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kfh[1], qh[ds], sacc[1], 0, 0, 0);
    }
    if constexpr (RAGGED) {
      const int lim = keys_left - 8 * kh;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (ks * 32 + 16 * (r >> 3) + (r & 7) >= lim) sacc[ks][r] = -INFINITY;
    }
  }
  FP_DEVICE void softmax() {
    // (fmaxf, not v_max3_f32 as inline asm: the hazard recognizer does not place the wait states an MFMA result needs in front of an asm reader)
    float mch[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {  // four chains of eight scores
      const f32x16& s = sacc[j >> 1];
      const int b = (j & 1) * 8;
      float t = fmaxf(fmaxf(s[b], s[b + 1]), s[b + 2]);
      t = fmaxf(fmaxf(t, s[b + 3]), s[b + 4]);
      t = fmaxf(fmaxf(t, s[b + 5]), s[b + 6]);
      mch[j] = fmaxf(t, s[b + 7]);
    }
    float mx = fmaxf(fmaxf(fmaxf(mch[0], mch[1]), mch[2]), mch[3]);
    {  // the query's other 32 scores live in lane ^ 32 (one v_permlane32_swap; results through temporaries, see attn_bf16_w64_kernel)
      const unsigned mu = __builtin_bit_cast(unsigned, mx);
      const auto sw = __builtin_amdgcn_permlane32_swap(mu, mu, false, false);
      const unsigned m0 = sw[0], m1 = sw[1];
      mx = fmaxf(__builtin_bit_cast(float, m0), __builtin_bit_cast(float, m1));
    }
    const bool moves = mx - m_run > SPLIT_LAZY_TH;
    const bool grow = __any(moves);
    float alpha = 1.f;
    if (grow) {
      const float m_new = moves ? mx : m_run;
      alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
      m_run = m_new;
    }
    const float mc = fmaf(m_run, c, -SPLIT_P_LOG2_SCALE);
    float ps[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = __builtin_amdgcn_exp2f(fmaf(sacc[ks][r], c, -mc));
        sacc[ks][r] = p;
        ps[r & 3] += p;
      }
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        unsigned h[4], l[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float pa = sacc[ks][8 * kk + 2 * e], pb = sacc[ks][8 * kk + 2 * e + 1];
          h[e] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{pa, pb}, f16x2));  // v_cvt_pk_f16_f32 (RNE)
          // lo = f16(p' - hi): the difference is exact in fp32, one rounding -- the bits of split16_pack2_inrange
          asm("v_fma_mixlo_f16 %0, -%1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(l[e]) : "v"(h[e]), "v"(pa));
          asm("v_fma_mixhi_f16 %0, -%1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l[e]) : "v"(h[e]), "v"(pb));
        }
        ph[ks * 2 + kk] = __builtin_bit_cast(f16x8, make_uint4(h[0], h[1], h[2], h[3]));
        pl[ks * 2 + kk] = __builtin_bit_cast(f16x8, make_uint4(l[0], l[1], l[2], l[3]));
      }
    }
    if (grow) {
      l_run *= alpha;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[i][r] *= alpha;
    }
    l_run += (ps[0] + ps[1]) + (ps[2] + ps[3]);
    asm volatile("" : "+v"(l_run));  // the row sum stays in this phase (the scheduler otherwise sinks its adds behind the next phase's MFMAs)
  }
  FP_DEVICE void pv(const char* Vs) {
#pragma unroll
    for (int kstep = 0; kstep < 4; ++kstep) {
      f16x8 vfh[2], vfl[2];
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        const char* vph = Vs + vrd0 + (((2 * dt) ^ kq) << 6) + kstep * 4096;       // hi unit of d-group dt (+ 1024 B = 4 keys on)
        const char* vpl = Vs + vrd0 + (((2 * dt + 1) ^ kq) << 6) + kstep * 4096;   // lo unit
#ifdef SPP_NO_LDS
        vfh[dt] = qh[(kstep + dt) & 3]; vfl[dt] = ql[(kstep + dt + 1) & 3];
        vfh[dt][0] = (_Float16)(float)((size_t)Vs & 0x8000 ? 1 : 2); vfl[dt][1] = (_Float16)(float)((size_t)Vs & 0x10000 ? 1 : 3);
        (void)vph; (void)vpl;
#else
        const s16x4 h_lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(vph));
        const s16x4 h_hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(vph + 1024));
        const s16x4 l_lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(vpl));
        const s16x4 l_hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(vpl + 1024));
        vfh[dt] = __builtin_bit_cast(f16x8, __builtin_shufflevector(h_lo, h_hi, 0, 1, 2, 3, 4, 5, 6, 7));
        vfl[dt] = __builtin_bit_cast(f16x8, __builtin_shufflevector(l_lo, l_hi, 0, 1, 2, 3, 4, 5, 6, 7));
#endif
      }
      oacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vfl[0], ph[kstep], oacc[0], 0, 0, 0);
      oacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vfl[1], ph[kstep], oacc[1], 0, 0, 0);
      oacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vfh[0], pl[kstep], oacc[0], 0, 0, 0);
      oacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vfh[1], pl[kstep], oacc[1], 0, 0, 0);
      oacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vfh[0], ph[kstep], oacc[0], 0, 0, 0);
      oacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vfh[1], ph[kstep], oacc[1], 0, 0, 0);
    }
  }
  // The matrix phase of the role-split kernel's steady state -- P V of one tile, then S^T of the next -- as one pinned instruction order: the
  // wave is alone on its SIMD's matrix pipe (its partner is in its softmax), so nobody else hides its LDS latency or fills its issue gaps.
  // Eight steps of six MFMAs (four P V steps of 16 keys, four S steps of 16 d); the fragments of step i + 1 are read in the gaps behind the
  // first four MFMAs of step i (two transpose reads or one b128 read per gap), `dma` (the wave's LDS-DMA loads of the tiles ahead) rides in the
  // gaps of the first S step -- behind the last transpose read of the iteration.  Same MFMAs, same order per accumulator as pv() + scores().
  template <class F>
  FP_DEVICE void matrix(const char* Vs, const char* Ks, F&& dma) {
#define SPP_PIN() __builtin_amdgcn_sched_barrier(0)
    // (Measurement switches SPP_NOP_A / SPP_NOP_B: `s_nop n` behind every MFMA of this phase, A: one followed by fragment reads, B: one that is not.
    // A wave whose NEXT instruction is an MFMA waiting for the busy matrix pipe holds the SIMD's issue arbitration -- beside 48 dense MFMAs a
    // partner's 64 v_fma take 1536 cycles instead of 340, with one s_nop 7 per MFMA 408 (tools/ubench/mfma_valu_corun.hip) -- but THIS phase has
    // fragment reads and their waits between its MFMAs already: every nop setting measured slower, 803 us without -> 807 ... 845 us.  Off.)
#ifndef SPP_NOP_A
#define SPP_NOP_A -1
#endif
#ifndef SPP_NOP_B
#define SPP_NOP_B -1
#endif
#define SPP_STR2(x) #x
#define SPP_STR(x) SPP_STR2(x)
#define SPP_GAP_A() do { if (SPP_NOP_A >= 0) asm volatile("s_nop " SPP_STR(SPP_NOP_A)); } while (0)
#define SPP_GAP_B() do { if (SPP_NOP_B >= 0) asm volatile("s_nop " SPP_STR(SPP_NOP_B)); } while (0)
    auto ldv = [&](int kstep, int part) {  // part: 0 = hi of d-group 0, 1 = lo of d-group 0, 2 = hi of d-group 1, 3 = lo of d-group 1
      const char* vp = Vs + vrd0 + ((part ^ kq) << 6) + kstep * 4096;
      const s16x4 x = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(vp));
      const s16x4 y = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(vp + 1024));
      return __builtin_bit_cast(f16x8, __builtin_shufflevector(x, y, 0, 1, 2, 3, 4, 5, 6, 7));
    };
    auto ldk = [&](int ds, int part) {     // part: 0 = hi of key half 0, 1 = lo of key half 0, 2 = hi of key half 1, 3 = lo of key half 1
      const int ch = (ds >> 1) * 8 + (ds & 1) * 2 + kh + (part & 1) * 4, row = (part >> 1) * 32 + krow;
      return *reinterpret_cast<const f16x8*>(Ks + row * 256 + ((ch ^ (row & 15)) << 4));
    };
    f16x8 c[4], n[4];
#pragma unroll
    for (int part = 0; part < 4; ++part) c[part] = ldv(0, part);
    SPP_PIN();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      oacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(c[1], ph[k], oacc[0], 0, 0, 0);
      SPP_PIN();
      n[0] = k < 3 ? ldv(k + 1, 0) : ldk(0, 0);
      SPP_GAP_A();
      SPP_PIN();
      oacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(c[3], ph[k], oacc[1], 0, 0, 0);
      SPP_PIN();
      n[1] = k < 3 ? ldv(k + 1, 1) : ldk(0, 1);
      SPP_GAP_A();
      SPP_PIN();
      oacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(c[0], pl[k], oacc[0], 0, 0, 0);
      SPP_PIN();
      n[2] = k < 3 ? ldv(k + 1, 2) : ldk(0, 2);
      SPP_GAP_A();
      SPP_PIN();
      oacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(c[2], pl[k], oacc[1], 0, 0, 0);
      SPP_PIN();
      n[3] = k < 3 ? ldv(k + 1, 3) : ldk(0, 3);
      SPP_GAP_A();
      SPP_PIN();
      oacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(c[0], ph[k], oacc[0], 0, 0, 0);
      SPP_PIN();
      SPP_GAP_B();
      SPP_PIN();
      oacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(c[2], ph[k], oacc[1], 0, 0, 0);
      SPP_PIN();
      SPP_GAP_B();
      SPP_PIN();
#pragma unroll
      for (int part = 0; part < 4; ++part) c[part] = n[part];
    }
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ds = 0; ds < 4; ++ds) {
      sacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(c[1], qh[ds], ds ? sacc[0] : zero, 0, 0, 0);
      SPP_PIN();
      if (ds < 3) n[0] = ldk(ds + 1, 0);
      SPP_GAP_A();
      SPP_PIN();
      sacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(c[3], qh[ds], ds ? sacc[1] : zero, 0, 0, 0);
      SPP_PIN();
      if (ds < 3) n[1] = ldk(ds + 1, 1);
      SPP_GAP_A();
      SPP_PIN();
      sacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(c[0], ql[ds], sacc[0], 0, 0, 0);
      SPP_PIN();
      if (ds < 3) n[2] = ldk(ds + 1, 2);
      SPP_GAP_A();
      SPP_PIN();
      sacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(c[2], ql[ds], sacc[1], 0, 0, 0);
      SPP_PIN();
      if (ds < 3) n[3] = ldk(ds + 1, 3);
      SPP_GAP_A();
      SPP_PIN();
      sacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(c[0], qh[ds], sacc[0], 0, 0, 0);
      SPP_PIN();
      if (ds == 0) dma();
      SPP_GAP_B();
      SPP_PIN();
      sacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(c[2], qh[ds], sacc[1], 0, 0, 0);
      SPP_PIN();
      SPP_GAP_B();
      SPP_PIN();
      if (ds < 3) {
#pragma unroll
        for (int part = 0; part < 4; ++part) c[part] = n[part];
      }
    }
#undef SPP_GAP_A
#undef SPP_GAP_B
#undef SPP_PIN
  }
};

// Epilogue of the split-fp16 attention kernels: O = sum(P v) / l without the scales of P and v, written as a split-fp16 row or an f16f8 row.
FP_DEVICE void split_attn_store(const AttnArgs& a, const f32x16 (&oacc)[2], float l_run, int q0, int l31, int kh, int NQ, int sel_base, int img, int N, int head) {
  const int q = q0 + l31;
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.f / (l_tot * a.in_scale);  // O = sum(P' v') / l': P' and l' carry the same 2^14, v' its operand scale
  // 16-byte stores: a lane's 4 consecutive d and lane ^ 32's next 4 paired by v_permlane32_swap (attn_bf16_w64_kernel's epilogue)
  const size_t orow = a.sel_off ? (size_t)(sel_base + q) : (size_t)img * N + q;  // compact rows in selected mode
  _Float16* o = reinterpret_cast<_Float16*>(a.out) + orow * a.ld_out + head * 128;
  // a convex combination of v rows that fit their scale cannot clamp -- but a NaN / Inf born inside the attention (an overflowing score)
  // would leave the v_med3 of the packing as a finite operand: the NaN-propagating running maximum is what reports it
  float o_amax = 0.f;
  if (a.out_fmt == 1) {
    // f16f8 row (common.hpp): a head's 64 output dims are one 64-column group -- fp16 high halves (128 B), then e4m3(hi) and e4m3(lo) (64 B
    // each).  Same lane exchange as below; the word that travels in the lo slot is [hi8_a, hi8_b, lo8_a, lo8_b] of a column pair.
    char* ob = reinterpret_cast<char*>(o);
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        unsigned xh0, xp0, xh1, xp1, yh0, yp0, yh1, yp1;
        splitx_pack2(oacc[dt][8 * j + 0] * inv, oacc[dt][8 * j + 1] * inv, a.out_scale, xh0, xp0, o_amax);
        splitx_pack2(oacc[dt][8 * j + 2] * inv, oacc[dt][8 * j + 3] * inv, a.out_scale, xh1, xp1, o_amax);
        splitx_pack2(oacc[dt][8 * j + 4] * inv, oacc[dt][8 * j + 5] * inv, a.out_scale, yh0, yp0, o_amax);
        splitx_pack2(oacc[dt][8 * j + 6] * inv, oacc[dt][8 * j + 7] * inv, a.out_scale, yh1, yp1, o_amax);
        const auto h0 = __builtin_amdgcn_permlane32_swap(xh0, yh0, false, false);
        const auto h1 = __builtin_amdgcn_permlane32_swap(xh1, yh1, false, false);
        const auto p0 = __builtin_amdgcn_permlane32_swap(xp0, yp0, false, false);
        const auto p1 = __builtin_amdgcn_permlane32_swap(xp1, yp1, false, false);
        const int d = dt * 32 + 16 * j + 8 * kh;   // this lane's 8 consecutive dims
        if (q < NQ) {
          *reinterpret_cast<uint4*>(ob + d * 2) = make_uint4(h0[0], h1[0], h0[1], h1[1]);
          *reinterpret_cast<uint2*>(ob + 128 + d) = make_uint2(__builtin_amdgcn_perm(p1[0], p0[0], 0x05040100u), __builtin_amdgcn_perm(p1[1], p0[1], 0x05040100u));
          *reinterpret_cast<uint2*>(ob + 192 + d) = make_uint2(__builtin_amdgcn_perm(p1[0], p0[0], 0x07060302u), __builtin_amdgcn_perm(p1[1], p0[1], 0x07060302u));
        }
      }
  } else {
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      unsigned xh0, xl0, xh1, xl1, yh0, yl0, yh1, yl1;
      split16_pack2(oacc[dt][8 * j + 0] * inv, oacc[dt][8 * j + 1] * inv, a.out_scale, xh0, xl0, o_amax);
      split16_pack2(oacc[dt][8 * j + 2] * inv, oacc[dt][8 * j + 3] * inv, a.out_scale, xh1, xl1, o_amax);
      split16_pack2(oacc[dt][8 * j + 4] * inv, oacc[dt][8 * j + 5] * inv, a.out_scale, yh0, yl0, o_amax);
      split16_pack2(oacc[dt][8 * j + 6] * inv, oacc[dt][8 * j + 7] * inv, a.out_scale, yh1, yl1, o_amax);
      const auto h0 = __builtin_amdgcn_permlane32_swap(xh0, yh0, false, false);
      const auto h1 = __builtin_amdgcn_permlane32_swap(xh1, yh1, false, false);
      const auto l0 = __builtin_amdgcn_permlane32_swap(xl0, yl0, false, false);
      const auto l1 = __builtin_amdgcn_permlane32_swap(xl1, yl1, false, false);
      _Float16* op = o + dt * 64 + 16 * j + 8 * kh;
      if (q < NQ) {
        *reinterpret_cast<uint4*>(op) = make_uint4(h0[0], h1[0], h0[1], h1[1]);
        *reinterpret_cast<uint4*>(op + 32) = make_uint4(l0[0], l1[0], l0[1], l1[1]);
      }
    }
  }
  if (q < NQ) report_saturation(a.sat, 0, o_amax, FP_F16_MAX);
}


__global__ __launch_bounds__(512, 2) void attn_split_kernel(AttnArgs a) {
  __shared__ __attribute__((aligned(16))) char KV[2][2][16384];  // [stage][K | V]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, kh = lane >> 5;
  int qt, head, img;
  {
    const int nqt = ((a.sel_off ? a.max_sel : a.n_tok) + 255) / 256, pairs = a.heads * a.batch, i = blockIdx.x;
    int pair;
    if ((pairs & 7) == 0) {  // all query tiles of an (image, head) pair on one XCD (one L2), see attn_bf16_kernel
      const int j = i >> 3;
      if (a.tail_last && nqt > 1) {  // ... the pair's last tile (short: half a block's work or less) at the END of the XCD's sequence
        const int nfull = (pairs >> 3) * (nqt - 1);
        const bool tail = j >= nfull;
        qt = tail ? nqt - 1 : j % (nqt - 1);
        pair = (tail ? j - nfull : j / (nqt - 1)) * 8 + (i & 7);
      } else {
        qt = j % nqt;
        pair = (j / nqt) * 8 + (i & 7);
      }
    } else {
      qt = i % nqt;
      pair = i / nqt;
    }
    head = pair % a.heads;
    img = pair / a.heads;
  }
  const int N = a.n_tok, D = a.dim;
  const _Float16* qkv = reinterpret_cast<const _Float16*>(a.qkv) + (size_t)img * N * a.ld_qkv;
  // queries: every token, or the image's selected tokens (keys / values: always all N tokens), as in attn_bf16_w64_kernel
  const int sel_base = a.sel_off ? a.sel_off[img] : 0;
  const int NQ = a.sel_off ? a.sel_off[img + 1] - sel_base : N;
  if (qt * 256 >= NQ) return;  // selected mode: the grid is sized for the image with the most queries (block-uniform, before any barrier)
  const int q0 = qt * 256 + wave * 32;
  const bool active = q0 < NQ;  // wave-uniform; an inactive wave only stages tiles and keeps the barriers

  // ---- staging: a DMA instruction moves 4 rows x 256 B; wave w issues row groups 2w, 2w + 1 of K and of V
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)qkv, 0, (unsigned)((size_t)N * a.ld_qkv * 2), 0x00020000);
  const int sr = lane >> 4, sp = lane & 15;
  const unsigned rowoff = (unsigned)(sr * a.ld_qkv) * 2u;
  // K: row r = 4 G + sr of the tile -> r & 15 = 4 (G & 3) + sr, with G = 2 wave (+ 1)
  const unsigned voff_k0 = rowoff + ((unsigned)(sp ^ (4 * ((2 * wave) & 3) + sr)) << 4);
  const unsigned voff_k1 = rowoff + ((unsigned)(sp ^ (4 * ((2 * wave + 1) & 3) + sr)) << 4);
  const unsigned voff_v = rowoff + ((unsigned)((((sp >> 2) ^ sr) << 2) | (sp & 3)) << 4);   // r & 3 = sr
  const unsigned tile_stride = (unsigned)(64 * a.ld_qkv) * 2u, grp_stride = (unsigned)(4 * a.ld_qkv) * 2u;
  const unsigned soff_k = (unsigned)(2 * D + head * 128) * 2u + 2u * wave * grp_stride, soff_v = soff_k + (unsigned)(2 * D) * 2u;
  auto stage_tile = [&](int kt, int stage) {
    const unsigned t = kt * tile_stride;
    char* kd = KV[stage][0] + wave * 2048;
    char* vd = KV[stage][1] + wave * 2048;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_t*)kd, 16, voff_k0, soff_k + t, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_t*)(kd + 1024), 16, voff_k1, soff_k + t + grp_stride, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_t*)vd, 16, voff_v, soff_v + t, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_t*)(vd + 1024), 16, voff_v, soff_v + t + grp_stride, 0, 0);
  };
  const int nkt = (N + 63) / 64;
  stage_tile(0, 0);

  SplitAttnWave w;
  w.init(lane, a.in_scale);
  {
    const int q = q0 + l31;
    int qc = q < NQ ? q : NQ - 1;
    if (a.sel_rows) qc = a.sel_rows[sel_base + qc] - img * N;  // the selected query's token
    w.load_q(qkv + (size_t)qc * a.ld_qkv + head * 128);
  }
  __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0) as a builtin (see attn_bf16_w64_kernel)
  __syncthreads();
  const int nfull = N / 64;
  for (int kt = 0; kt < nfull; ++kt) {
    if (kt + 1 < nkt) stage_tile(kt + 1, (kt + 1) & 1);
    if (active) {
      w.scores<false>(KV[kt & 1][0], 64);
      w.softmax();
      w.pv(KV[kt & 1][1]);
    }
    __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0): the compiler does not wait for LDS-DMA before a barrier
    __syncthreads();
  }
  if (nfull < nkt && active) {  // the ragged last tile
    w.scores<true>(KV[nfull & 1][0], N - nfull * 64);
    w.softmax();
    w.pv(KV[nfull & 1][1]);
  }
  if (active) split_attn_store(a, w.oacc, w.l_run, q0, l31, kh, NQ, sel_base, img, N, head);
}

#ifdef FP_EXPERIMENTS   // built, bit-identical and ~3 % SLOWER than attn_split_kernel (profiles/EXPERIMENTS.md round 5): measurement builds only
// ---------------------------------------------------------------- split-fp16, role-split ("ping-pong") form (AttnArgs.variant 2 of the split kernels)
// The same arithmetic as attn_split_kernel, instruction for instruction per query (same MFMA order per accumulator, same softmax, same
// epilogue: bit-identical outputs), on another schedule.  In the lock-step kernel the two waves a 512-thread workgroup places on each SIMD
// (waves w and w + 4) walk S -> softmax -> P V together behind one barrier per key tile, so the SIMD's matrix pipe idles while both waves are
// in their softmax and its VALU while both are in their MFMAs: a tile costs the SUM of the two (measured: 48 MFMAs = 1536 cycles + ~300 VALU
// per wave and tile, 7.0 k cycles per tile for the pair).  Here the waves of a SIMD run half a tile apart:
//   even interval 2t:   waves 0-3: P V (t-1), S(t)         [matrix pipe]      waves 4-7: softmax(t-1)               [VALU]
//   odd interval 2t+1:  waves 0-3: softmax(t)              [VALU]             waves 4-7: P V (t-1), S(t)           [matrix pipe]
// with a barrier at the end of every interval (the one after the even interval orders nothing in memory, it keeps the two halves in anti-phase).
// K / V tiles travel through a ring of three slots: every wave issues its rows of K(t+2) and V(t+1) right behind its P V (t-1) -- after its last
// transpose read of the iteration (the compiler drains vmcnt in front of the first transpose read that follows an LDS-DMA) --, the loads of
// iteration t have landed when iteration t + 1 ends (vmcnt(4) + barrier), one iteration ahead of their first reader.
//   slot s: [K image 16 KiB | V image 16 KiB] (layouts as in attn_split_kernel); K(t) and V(t) live in slot t % 3
// A short last query tile hands its 32-query blocks to waves 0, 4, 1, 5, ... so that both halves of a SIMD have work.
constexpr int SPP_SLOT = 32768, SPP_LDS = 3 * SPP_SLOT;
#ifdef SPP_NO_MIDBAR   // (measurement build: without the barrier that holds the two halves in anti-phase)
#define SPP_MID_BARRIER() ((void)0)
#else
#define SPP_MID_BARRIER() __builtin_amdgcn_s_barrier()
#endif

__global__ __launch_bounds__(512, 2) void attn_split_pp_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char KVR[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, kh = lane >> 5;
#if defined(SPP_GRP) && SPP_GRP == 1   // (measurement builds: tools/attn_split_ablate.sh)
  const int grp = wave & 1;
#else
  const int grp = wave >> 2;  // 0: matrix phase in the even intervals; 1: half a tile behind
#endif
  int qt, head, img;
  {
    const int nqt = ((a.sel_off ? a.max_sel : a.n_tok) + 255) / 256, pairs = a.heads * a.batch, i = blockIdx.x;
    int pair;
    if ((pairs & 7) == 0) {  // as in attn_split_kernel: a pair's query tiles on one XCD, the short last tile at the end of the XCD's sequence
      const int j = i >> 3;
      if (a.tail_last && nqt > 1) {
        const int nfull = (pairs >> 3) * (nqt - 1);
        const bool tail = j >= nfull;
        qt = tail ? nqt - 1 : j % (nqt - 1);
        pair = (tail ? j - nfull : j / (nqt - 1)) * 8 + (i & 7);
      } else {
        qt = j % nqt;
        pair = (j / nqt) * 8 + (i & 7);
      }
    } else {
      qt = i % nqt;
      pair = i / nqt;
    }
    head = pair % a.heads;
    img = pair / a.heads;
  }
  const int N = a.n_tok, D = a.dim;
  const _Float16* qkv = reinterpret_cast<const _Float16*>(a.qkv) + (size_t)img * N * a.ld_qkv;
  const int sel_base = a.sel_off ? a.sel_off[img] : 0;
  const int NQ = a.sel_off ? a.sel_off[img + 1] - sel_base : N;
  if (qt * 256 >= NQ) return;  // block-uniform, before any barrier
#if defined(SPP_GRP) && SPP_GRP == 1
  const int q0 = qt * 256 + wave * 32;
#else
  const int q0 = qt * 256 + (((wave & 3) << 1) | grp) * 32;
#endif
  const bool active = q0 < NQ;  // wave-uniform; an inactive wave only stages tiles and keeps the barriers

  // ---- staging (attn_split_kernel's): a DMA instruction moves 4 rows x 256 B; wave w issues row groups 2w, 2w + 1 of K and of V
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)qkv, 0, (unsigned)((size_t)N * a.ld_qkv * 2), 0x00020000);
  const int sr = lane >> 4, sp = lane & 15;
  const unsigned rowoff = (unsigned)(sr * a.ld_qkv) * 2u;
  const unsigned voff_k0 = rowoff + ((unsigned)(sp ^ (4 * ((2 * wave) & 3) + sr)) << 4);
  const unsigned voff_k1 = rowoff + ((unsigned)(sp ^ (4 * ((2 * wave + 1) & 3) + sr)) << 4);
  const unsigned voff_v = rowoff + ((unsigned)((((sp >> 2) ^ sr) << 2) | (sp & 3)) << 4);
  const unsigned tile_stride = (unsigned)(64 * a.ld_qkv) * 2u, grp_stride = (unsigned)(4 * a.ld_qkv) * 2u;
  const unsigned soff_k = (unsigned)(2 * D + head * 128) * 2u + 2u * wave * grp_stride, soff_v = soff_k + (unsigned)(2 * D) * 2u;
  auto stage_k = [&](int kt, int slot) {
    const unsigned t = kt * tile_stride;
    char* kd = KVR + slot * SPP_SLOT + wave * 2048;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_t*)kd, 16, voff_k0, soff_k + t, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_t*)(kd + 1024), 16, voff_k1, soff_k + t + grp_stride, 0, 0);
  };
  auto stage_v = [&](int kt, int slot) {
    const unsigned t = kt * tile_stride;
    char* vd = KVR + slot * SPP_SLOT + 16384 + wave * 2048;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_t*)vd, 16, voff_v, soff_v + t, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_t*)(vd + 1024), 16, voff_v, soff_v + t + grp_stride, 0, 0);
  };
  const int T = (N + 63) / 64;
  stage_k(0, 0);
  stage_v(0, 0);
  if (T > 1) stage_k(1, 1);

  SplitAttnWave w;
  w.init(lane, a.in_scale);
  {
    const int q = q0 + l31;
    int qc = q < NQ ? q : NQ - 1;
    if (a.sel_rows) qc = a.sel_rows[sel_base + qc] - img * N;
    w.load_q(qkv + (size_t)qc * a.ld_qkv + head * 128);
  }
#ifdef SPP_NO_MATRIX   // measurement build (tools/attn_split_ablate.sh): no MFMAs, scores = raw LDS words (values the compiler cannot fold)
  auto s_phase = [&](int t, int slot) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int r = 0; r < 16; r += 4) {
        const float4 x = *reinterpret_cast<const float4*>(KVR + slot * SPP_SLOT + (ks * 32 + l31) * 256 + kh * 64 + r * 4);
        w.sacc[ks][r] = x.x * 1e-30f; w.sacc[ks][r + 1] = x.y * 1e-30f; w.sacc[ks][r + 2] = x.z * 1e-30f; w.sacc[ks][r + 3] = x.w * 1e-30f;
      }
  };
  auto pv_phase = [&](int slot) {
#pragma unroll
    for (int r = 0; r < 16; ++r) w.oacc[0][r] += __builtin_bit_cast(float, __builtin_bit_cast(uint4, w.ph[r & 3])[r >> 2]) * 1e-30f;
  };
#else
  auto s_phase = [&](int t, int slot) {
    if (N - t * 64 < 64) w.scores<true>(KVR + slot * SPP_SLOT, N - t * 64);   // only the last tile can be ragged
    else w.scores<false>(KVR + slot * SPP_SLOT, 64);
  };
  auto pv_phase = [&](int slot) { w.pv(KVR + slot * SPP_SLOT + 16384); };
#endif
#ifdef SPP_NO_SOFTMAX  // measurement build: the matrix phases alone (the scores stay live: no MFMA is eliminated)
  auto softmax_phase = [&]() {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int r = 0; r < 16; ++r) asm volatile("" : "+v"(w.sacc[ks][r]));
  };
#elif defined(SPP_PRIO) && SPP_PRIO == 1   // measurement builds: the softmax phase / the matrix phase at raised issue priority
  auto softmax_phase = [&]() { __builtin_amdgcn_s_setprio(3); w.softmax(); __builtin_amdgcn_s_setprio(0); };
#else
  auto softmax_phase = [&]() { w.softmax(); };
#endif
  // this wave's rows of K(t+2) and V(t+1): always four loads (a tile index past the end reads out of range: zeros into a slot nobody reads),
  // so "everything but this iteration's loads has landed" is vmcnt(4) in every iteration
  auto stage_ahead = [&](int t, int s_prev, int s_next) {
    stage_k(t + 2 < T ? t + 2 : T, s_prev);
    stage_v(t + 1 < T ? t + 1 : T, s_next);
  };
#if defined(SPP_NO_MATRIX) || defined(SPP_NO_LDS) || defined(SPP_NO_PIPE)
  auto matrix_phase = [&](int t, int sp, int sc, int sn) { pv_phase(sp); stage_ahead(t, sp, sn); s_phase(t, sc); };
#else
  auto matrix_phase = [&](int t, int sp, int sc, int sn) {
#if defined(SPP_PRIO) && SPP_PRIO == 2
    __builtin_amdgcn_s_setprio(3);
#endif
    w.matrix(KVR + sp * SPP_SLOT + 16384, KVR + sc * SPP_SLOT, [&]() { stage_ahead(t, sp, sn); });
#if defined(SPP_PRIO) && SPP_PRIO == 2
    __builtin_amdgcn_s_setprio(0);
#endif
  };
#endif
  auto end_of_iteration = [&]() {
    __builtin_amdgcn_s_waitcnt(0x0f74);  // vmcnt(4)
    __builtin_amdgcn_s_barrier();
  };
  auto next = [](int& s_prev, int& s_cur, int& s_next) { s_prev = s_cur, s_cur = s_next, s_next = s_next == 2 ? 0 : s_next + 1; };

  __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0) as a builtin (see attn_bf16_w64_kernel)
  __syncthreads();
  int s_prev = 2, s_cur = 0, s_next = 1;
  if (!active) {  // a wave without queries (short last query tile): its share of the staging and the barriers
    for (int t = 0; t < T; ++t) {
      if (grp == 0) stage_ahead(t, s_prev, s_next);
      SPP_MID_BARRIER();
      if (grp != 0) stage_ahead(t, s_prev, s_next);
      end_of_iteration();
      next(s_prev, s_cur, s_next);
    }
    SPP_MID_BARRIER();
    return;
  }
#ifdef SPP_TIMELINE   // measurement build (tools/spp_timeline.py): s_memtime stamps of one workgroup's steady-state iterations, waves 0 and 4
  long long* tl = reinterpret_cast<long long*>(KVR + SPP_LDS) + wave * 64;
  int tl_n = 0;
  const bool tl_on = blockIdx.x == SPP_TIMELINE && (wave & 3) == 0;
#define SPP_STAMP() do { if (tl_on && tl_n < 64) { if (lane == 0) tl[tl_n] = clock64(); ++tl_n; } } while (0)
#else
#define SPP_STAMP() ((void)0)
#endif
  if (grp == 0) {
    stage_ahead(0, s_prev, s_next);
    s_phase(0, s_cur);
    SPP_MID_BARRIER();
    softmax_phase();
    end_of_iteration();
    next(s_prev, s_cur, s_next);
    for (int t = 1; t < T; ++t) {
      SPP_STAMP();
      if (t < T - 1) {
        matrix_phase(t, s_prev, s_cur, s_next);
      } else {
        pv_phase(s_prev);
        stage_ahead(t, s_prev, s_next);
        s_phase(t, s_cur);
      }
      SPP_STAMP();
      SPP_MID_BARRIER();
      SPP_STAMP();
      softmax_phase();
      SPP_STAMP();
      __builtin_amdgcn_s_waitcnt(0x0f74);
      SPP_STAMP();
      __builtin_amdgcn_s_barrier();
      next(s_prev, s_cur, s_next);
    }
    pv_phase(s_prev);
    SPP_MID_BARRIER();
  } else {
    SPP_MID_BARRIER();
    stage_ahead(0, s_prev, s_next);
    s_phase(0, s_cur);
    end_of_iteration();
    next(s_prev, s_cur, s_next);
    for (int t = 1; t < T; ++t) {
      SPP_STAMP();
      softmax_phase();   // of tile t - 1
      SPP_STAMP();
      SPP_MID_BARRIER();
      SPP_STAMP();
      if (t < T - 1) {
        matrix_phase(t, s_prev, s_cur, s_next);
      } else {
        pv_phase(s_prev);
        stage_ahead(t, s_prev, s_next);
        s_phase(t, s_cur);
      }
      SPP_STAMP();
      __builtin_amdgcn_s_waitcnt(0x0f74);
      SPP_STAMP();
      __builtin_amdgcn_s_barrier();
      next(s_prev, s_cur, s_next);
    }
    softmax_phase();
    SPP_MID_BARRIER();
    pv_phase(s_prev);
  }
#ifdef SPP_TIMELINE
  if (tl_on && lane == 0) {
    long long* g = reinterpret_cast<long long*>(a.sat);   // the measurement build borrows the saturation pointer: [2][64] stamps
    for (int i = 0; i < 64; ++i) g[(wave >> 2) * 64 + i] = i < tl_n ? tl[i] : 0;
  }
#endif
  if (active) split_attn_store(a, w.oacc, w.l_run, q0, l31, kh, NQ, sel_base, img, N, head);
}
#endif  // FP_EXPERIMENTS

// ---------------------------------------------------------------- fp32 parity-mode attention
// qkv fp32 [B*N, 3D]; one thread per query row; keys/values of the (image, head) streamed through LDS.
__global__ __launch_bounds__(256) void attn_f32_kernel(AttnArgs a) {
  __shared__ float Ks[64][64];
  __shared__ float Vs[64][64];
  const int tid = threadIdx.x;
  const int qt = blockIdx.x, head = blockIdx.y, img = blockIdx.z;
  const int N = a.n_tok, D = a.dim;
  const float* qkv = reinterpret_cast<const float*>(a.qkv) + (size_t)img * N * a.ld_qkv;
  const int q = qt * 256 + tid;
  const int qc = q < N ? q : N - 1;
  float qv[64], acc[64];
#pragma unroll
  for (int d = 0; d < 64; ++d) {
    qv[d] = qkv[(size_t)qc * a.ld_qkv + head * 64 + d] * 0.125f;  // q * scale, like upstream
    acc[d] = 0.f;
  }
  float m_run = -INFINITY, l_run = 0.f;
  for (int key0 = 0; key0 < N; key0 += 64) {
    __syncthreads();
    for (int e = tid; e < 64 * 64; e += 256) {
      const int r = e >> 6, d = e & 63;
      int key = key0 + r;
      key = key < N ? key : N - 1;
      Ks[r][d] = qkv[(size_t)key * a.ld_qkv + D + head * 64 + d];
      Vs[r][d] = qkv[(size_t)key * a.ld_qkv + 2 * D + head * 64 + d];
    }
    __syncthreads();
    const int cnt = min(64, N - key0);
    for (int r = 0; r < cnt; ++r) {
      float s = 0.f;
#pragma unroll
      for (int d = 0; d < 64; ++d) s = fmaf(qv[d], Ks[r][d], s);
      const float m_new = fmaxf(m_run, s);
      const float alpha = expf(m_run - m_new);
      const float p = expf(s - m_new);
      l_run = l_run * alpha + p;
#pragma unroll
      for (int d = 0; d < 64; ++d) acc[d] = fmaf(p, Vs[r][d], acc[d] * alpha);
      m_run = m_new;
    }
  }
  if (q < N) {
    float* o = reinterpret_cast<float*>(a.out) + ((size_t)img * N + q) * a.ld_out + head * 64;
    const float inv = 1.f / l_run;
#pragma unroll
    for (int d = 0; d < 64; ++d) o[d] = acc[d] * inv;
  }
}


// ---------------------------------------------------------------- fp32 mode on the fp32 MFMA (v_mfma_f32_32x32x2_f32)
// The exact-fp32 mode's attention as a flash kernel on the matrix pipe: every product and every accumulation is fp32 (the MFMA's fma
// chain), the softmax is the fp32 online softmax of the kernels above.  4 waves x one 32-query block per workgroup; a 64-key K / V tile
// lives in LDS as fp32 rows of 64 + 4 floats (272 B: eight consecutive rows start in the eight 16-B bank groups, so the b128 fragment
// reads are conflict free), filled through registers -- the next tile's global loads are issued before the current tile's MFMAs.
//   S^T = K Q^T : A = K (32 keys x 2 d), B = Q^T (2 d x 32 queries); the k-pair of step s is d = (s, 32 + s): lane (l31, kh) holds
//                 Q[query l31][32 kh + s] (pre-multiplied by head_dim^-0.5, exact) and reads K[key][32 kh + 0..31] as eight b128.
//   O^T += V^T P^T : the B operand of the step that multiplies keys (x, x + 4) is the score register itself -- a lane's register r of
//                 key half ks is key 32 ks + (r & 3) + 8 (r >> 2) + 4 kh, i.e. exactly (query l31, k = kh) -- no conversion, no
//                 exchange; A = V[that key][l31 + 32 dt], one b32 read per MFMA.
// 128 MFMAs of 64 cycles per wave and key tile: matrix-pipe bound (256 flop/clk/CU).
__global__ __launch_bounds__(256, 2) void attn_f32_mfma_kernel(AttnArgs a) {
  constexpr int LDR = 68;  // floats per LDS row
  __shared__ __attribute__((aligned(16))) float Ks[64 * LDR];
  __shared__ __attribute__((aligned(16))) float Vs[64 * LDR];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, kh = lane >> 5;
  const int qt = blockIdx.x, head = blockIdx.y, img = blockIdx.z;
  const int N = a.n_tok, D = a.dim;
  const float* qkv = reinterpret_cast<const float*>(a.qkv) + (size_t)img * N * a.ld_qkv;
  const int q = qt * 128 + wave * 32 + l31;
  const int qc = q < N ? q : N - 1;
  float qv[32];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float4 t = *reinterpret_cast<const float4*>(qkv + (size_t)qc * a.ld_qkv + head * 64 + 32 * kh + 4 * j);
    qv[4 * j + 0] = t.x * 0.125f; qv[4 * j + 1] = t.y * 0.125f; qv[4 * j + 2] = t.z * 0.125f; qv[4 * j + 3] = t.w * 0.125f;  // q * scale, like upstream
  }
  f32x16 oacc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[i][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  // staging: thread t moves float4 chunk (row = c >> 4, col4 = c & 15), c = t + 256 i, of K and of V
  float4 kst[4], vst[4];
  auto load_tile = [&](int key0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = tid + 256 * i, r = c >> 4, c4 = c & 15;
      int key = key0 + r;
      key = key < N ? key : N - 1;
      const float* src = qkv + (size_t)key * a.ld_qkv + D + head * 64 + 4 * c4;
      kst[i] = *reinterpret_cast<const float4*>(src);
      vst[i] = *reinterpret_cast<const float4*>(src + D);
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = tid + 256 * i, r = c >> 4, c4 = c & 15;
      *reinterpret_cast<float4*>(Ks + r * LDR + 4 * c4) = kst[i];
      *reinterpret_cast<float4*>(Vs + r * LDR + 4 * c4) = vst[i];
    }
  };
  const int nkt = (N + 63) / 64;
  load_tile(0);
  store_tile();
  __syncthreads();
  constexpr float LOG2E = 1.44269504088896340736f;
  for (int kt = 0; kt < nkt; ++kt) {
    const int key0 = kt * 64;
    if (kt + 1 < nkt) load_tile(key0 + 64);  // lands under this tile's MFMAs
    // ---- S^T = K Q^T
    f32x16 sacc[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[ks][r] = 0.f;
      float4 kf[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) kf[j] = *reinterpret_cast<const float4*>(Ks + (ks * 32 + l31) * LDR + 32 * kh + 4 * j);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        sacc[ks] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[j].x, qv[4 * j + 0], sacc[ks], 0, 0, 0);
        sacc[ks] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[j].y, qv[4 * j + 1], sacc[ks], 0, 0, 0);
        sacc[ks] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[j].z, qv[4 * j + 2], sacc[ks], 0, 0, 0);
        sacc[ks] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[j].w, qv[4 * j + 3], sacc[ks], 0, 0, 0);
      }
    }
    if (key0 + 64 > N) {  // ragged last tile: mask the padded keys
      const int lim = N - key0 - 4 * kh;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (ks * 32 + (r & 3) + 8 * (r >> 2) >= lim) sacc[ks][r] = -INFINITY;
    }
    // ---- online softmax (a query's 64 scores live in lanes l31 and l31 + 32); exact running maximum
    float mx = fmaxf(sacc[0][0], sacc[1][0]);
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(fmaxf(mx, sacc[0][r]), sacc[1][r]);
    {
      const unsigned mu = __builtin_bit_cast(unsigned, mx);
      const auto sw = __builtin_amdgcn_permlane32_swap(mu, mu, false, false);
      const unsigned m0 = sw[0], m1 = sw[1];
      mx = fmaxf(__builtin_bit_cast(float, m0), __builtin_bit_cast(float, m1));
    }
    const float m_new = fmaxf(m_run, mx);
    const bool grow = __any(m_new > m_run);
    float alpha = 1.f;
    if (grow) alpha = __builtin_amdgcn_exp2f((m_run - m_new) * LOG2E);
    m_run = m_new;
    const float mc = m_new * LOG2E;
    float psum = 0.f;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = __builtin_amdgcn_exp2f(fmaf(sacc[ks][r], LOG2E, -mc));
        sacc[ks][r] = p;
        psum += p;
      }
    if (grow) {
      l_run *= alpha;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[i][r] *= alpha;
    }
    l_run += psum;
    // ---- O^T += V^T P^T
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float* vrow = Vs + (ks * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh) * LDR + l31;
        oacc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(vrow[0], sacc[ks][r], oacc[0], 0, 0, 0);
        oacc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(vrow[32], sacc[ks][r], oacc[1], 0, 0, 0);
      }
    __syncthreads();  // every wave is done with the tile
    if (kt + 1 < nkt) store_tile();
    __syncthreads();
  }
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.f / l_tot;
  if (q < N) {
    float* o = reinterpret_cast<float*>(a.out) + ((size_t)img * N + q) * a.ld_out + head * 64;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4*>(o + dt * 32 + 8 * g + 4 * kh) =
            make_float4(oacc[dt][4 * g + 0] * inv, oacc[dt][4 * g + 1] * inv, oacc[dt][4 * g + 2] * inv, oacc[dt][4 * g + 3] * inv);
  }
}

}  // namespace

#ifdef SPP_TIMELINE
static long long* spp_tl_buf = nullptr;
extern "C" int fp_debug_spp_timeline(long long* host_out) {  // [2][64]: wave 0's and wave 4's stamps of the last role-split launch
  if (!spp_tl_buf) return 1;
  return hipMemcpy(host_out, spp_tl_buf, 2 * 64 * sizeof(long long), hipMemcpyDeviceToHost) == hipSuccess ? 0 : 3;
}
#endif
int attn_launch(const AttnArgs& a_in, int dtype, hipStream_t st) {
  AttnArgs a = a_in;
  FP_REQUIRE(a.dim % 64 == 0 && a.heads * 64 == a.dim, "attention: head_dim must be 64 (dim %d heads %d)", a.dim, a.heads);
  FP_REQUIRE(a.n_tok >= 1 && a.batch >= 1, "attention: empty problem");
  if (dtype == FP_DTYPE_BF16 || dtype == FP_DTYPE_F16) {
    const bool h16 = dtype == FP_DTYPE_F16;   // IEEE fp16 q | k | v and output (the "f16" mode): the default work split only
    FP_REQUIRE(!h16 || (a.variant == 0 && a.out_fp8_scale <= 0.f && (size_t)a.n_tok * a.ld_qkv * 2 < 0xffffffffull),
               "attention(f16): the default kernel only (no work-split variants, no fp8 output), one image's qkv rows within 4 GiB");
    FP_REQUIRE(a.ld_qkv % 8 == 0 && (a.out_fp8_scale > 0.f ? a.ld_out % 4 == 0 : a.ld_out % 8 == 0), "attention(bf16): leading dims must keep 16-byte alignment");
#ifdef FP_EXPERIMENTS
    FP_REQUIRE(a.variant >= 0 && a.variant <= 4, "attention: unknown kernel variant %d", a.variant);
#else   // the shipped library: the default work split and its 32-queries-per-wave cross-check; 2, 3, 4 (measured slower) live in FP_EXPERIMENTS builds
    FP_REQUIRE(a.variant == 0 || a.variant == 1, "attention: kernel variant %d exists in FP_EXPERIMENTS builds only (0 = default, 1 = the cross-check kernel)", a.variant);
#endif
#ifdef FP_ATTN_DEFAULT_VARIANT  // (measurement build for same-box A/B runs of the whole pipeline: that split where 0 was asked for)
    const int variant = a.variant == 0 ? FP_ATTN_DEFAULT_VARIANT : a.variant;
#else
    const int variant = a.variant;
#endif
    const int w64 = variant == 1 ? 0 : (variant == 2 ? 2 : 1);  // default: 64 queries per wave; 3: the same with 8 waves = 512-query blocks
    FP_REQUIRE(a.out_fp8_scale <= 0.f || (w64 && a.ld_out % 4 == 0), "attention: the fp8 output exists in the 64-queries-per-wave kernel only");
    const bool sel = a.sel_off != nullptr;
    FP_REQUIRE(!sel || (a.sel_rows && a.max_sel >= 1), "attention: query selection needs sel_rows, sel_off and max_sel >= 1");
    FP_REQUIRE(!sel || (w64 && (size_t)a.n_tok * a.ld_qkv * 2 < 0xffffffffull), "attention: query selection exists in the 64-queries-per-wave kernel only");
    if (w64 && (size_t)a.n_tok * a.ld_qkv * 2 < 0xffffffffull) {
      const unsigned grid = (unsigned)(cdiv(sel ? a.max_sel : a.n_tok, variant == 3 ? 512 : 256) * a.heads * a.batch);
      // Block order: the last query tile of an (image, head) pair is short (1374 tokens = 5 x 256 + 94: the QC = 1 tail block ends in about half
      // the time).  Interleaved with the full blocks the short ones leave the final round of the launch as long as a full block; at the END of
      // each XCD's sequence the launch drains through half-length blocks: 343.6 -> 335.9 us in isolation, pipeline 1107.2 vs 1101.1 detections/s
      // (same box, three alternations; AttnArgs.tail_last = 0 is the other order).
      a.tail_last = 1;
#ifdef FP_EXPERIMENTS
      if (variant == 3) hipLaunchKernelGGL((attn_bf16_w64_kernel<2, 8>), dim3(grid), dim3(512), 0, st, a);
      else if (variant == 4) hipLaunchKernelGGL((attn_bf16_w64_kernel<2, 4, true>), dim3(grid), dim3(256), 0, st, a);
      else if (w64 == 2) hipLaunchKernelGGL(attn_bf16_w64_kernel<1>, dim3(grid), dim3(512), 0, st, a);
      else
#endif
      if (h16) hipLaunchKernelGGL((attn_bf16_w64_kernel<2, 4, false, true>), dim3(grid), dim3(256), 0, st, a);
      else hipLaunchKernelGGL(attn_bf16_w64_kernel<2>, dim3(grid), dim3(256), 0, st, a);
    } else {
      hipLaunchKernelGGL(attn_bf16_kernel, dim3(cdiv(a.n_tok, 128) * a.heads * a.batch), dim3(256), 0, st, a);
    }
  } else if (dtype == FP_DTYPE_F16X3) {
    FP_REQUIRE(!a.sel_off || (a.sel_rows && a.max_sel >= 1), "attention: query selection needs sel_rows, sel_off and max_sel >= 1");
    FP_REQUIRE(a.ld_qkv % 8 == 0 && a.ld_qkv >= 6 * a.dim && a.ld_out % 8 == 0 && a.ld_out >= 2 * a.dim, "attention(f16x3): rows are split-fp16 (6D / 2D halves), 16-byte aligned");
    FP_REQUIRE(a.in_scale >= 1.f && a.out_scale > 0.f, "attention(f16x3): in_scale must be >= 1 (SPLIT_LAZY_TH is a constant in raw score units; p' = 2^14 p must stay <= 2^15) and out_scale positive");
    FP_REQUIRE((size_t)a.n_tok * a.ld_qkv * 2 < 0xffffffffull, "attention(f16x3): one image's qkv rows must fit a 4-GiB buffer resource");
    a.tail_last = 1;
    const dim3 grid((unsigned)(cdiv(a.sel_off ? a.max_sel : a.n_tok, 256) * a.heads * a.batch));
    // variant 0 / 1 = the lock-step kernel (what the pipeline runs), 2 = the role-split kernel (attn_split_pp_kernel's header; bit-identical, FP_EXPERIMENTS builds)
    if (a.variant != 2) {
      hipLaunchKernelGGL(attn_split_kernel, grid, dim3(512), 0, st, a);
    } else {
#ifndef FP_EXPERIMENTS
      fp_set_error("attention(f16x3): the role-split kernel (variant 2) exists in FP_EXPERIMENTS builds only");
      return FP_ERR_UNSUPPORTED;
#else
      static FpDeviceOnce once;
#ifndef SPP_TIMELINE
      fp_allow_dynamic_lds(once, attn_split_pp_kernel, SPP_LDS);
#endif
#ifdef SPP_TIMELINE
      if (!spp_tl_buf) (void)hipMalloc(&spp_tl_buf, 2 * 64 * sizeof(long long));
      a.sat = reinterpret_cast<int*>(spp_tl_buf);
      fp_allow_dynamic_lds(once, attn_split_pp_kernel, SPP_LDS + 4096);
      hipLaunchKernelGGL(attn_split_pp_kernel, grid, dim3(512), SPP_LDS + 4096, st, a);
#else
      hipLaunchKernelGGL(attn_split_pp_kernel, grid, dim3(512), SPP_LDS, st, a);
#endif
#endif  // FP_EXPERIMENTS
    }
  } else if (dtype == FP_DTYPE_F32) {
    FP_REQUIRE(!a.sel_off, "attention: query selection exists in the bf16 and f16x3 kernels");
    if (a.variant == 1) {  // the thread-per-query VALU kernel (one fma chain per score, keys in order): the cross-check of the MFMA kernel
      hipLaunchKernelGGL(attn_f32_kernel, dim3(cdiv(a.n_tok, 256), a.heads, a.batch), dim3(256), 0, st, a);
    } else {
      FP_REQUIRE(a.ld_qkv % 4 == 0 && a.ld_out % 4 == 0, "attention(fp32): leading dims must keep 16-byte alignment");
      hipLaunchKernelGGL(attn_f32_mfma_kernel, dim3(cdiv(a.n_tok, 128), a.heads, a.batch), dim3(256), 0, st, a);
    }
  } else {
    fp_set_error("attention: unsupported dtype %d", dtype);
    return FP_ERR_UNSUPPORTED;
  }
  FP_CHECK_LAUNCH("attention");
  return FP_OK;
}
