// bf16 MFMA GEMM for the ViT linear layers: C = A[M,K] * W[N,K]^T (+ fused epilogue), fp32 accumulate.
//
// This is the arithmetic the reference delegates to the DINOv2 backbone's nn.Linear layers
// (call site /root/reference/utils/dinov2_utils.py:257).  MI355X design:
//   * tile 128x128x64, 256 threads = 2x2 waves, each wave 64x64 = 2x2 v_mfma_f32_32x32x16_bf16
//   * A and W tiles go HBM -> LDS with global_load_lds (16 B/lane, no VGPR round trip), double buffered,
//     one barrier per K-tile, next tile's DMA in flight under the MFMAs
//   * LDS image is row-major [row][64 bf16]; bank conflicts of the ds_read_b128 fragment reads are
//     removed by XOR-swizzling the 16-B chunk index with (row>>1)&7 -- applied on the *source* address
//     (the DMA destination is lane-linear) and again on the read (guide section 5.4 rule 21)
//   * operands are fed to the MFMA swapped (W as the "A" operand) so each lane ends up with 4 consecutive
//     output columns of one row: 8-byte bf16 / 16-byte fp32 epilogue accesses
//   * blockIdx is remapped so that each XCD's L2 sees a contiguous run of tiles sharing A panels.
#include <cstdlib>

#include "common.hpp"
#include "kernels.hpp"

namespace {

constexpr int BK = 64;

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_cvoid;

FP_DEVICE int swz(int row) { return (row >> 1) & 7; }

// One DMA instruction: 8 rows x 64 bf16 (1 KiB) of a tile, row group `rblk`, into the lane-linear LDS image.
FP_DEVICE void stage_rows(const __bf16* __restrict__ g, int ld, int row0, int k0, char* lds, int rblk, int lane) {
  const int row = rblk * 8 + (lane >> 3);
  const int chunk = (lane & 7) ^ swz(row);  // logical 16-B chunk that must land at physical slot lane&7
  const __bf16* src = g + (size_t)(row0 + row) * ld + k0 + chunk * 8;
  __builtin_amdgcn_global_load_lds((gbl_cvoid*)src, (lds_void*)(lds + rblk * 1024), 16, 0, 0);
}

FP_DEVICE bf16x8 read_frag(const char* lds, int row, int chunk) {
  return *reinterpret_cast<const bf16x8*>(lds + row * 128 + ((chunk ^ swz(row)) << 4));
}

// GELU(x) = 0.5 x (1 + erf(x / sqrt 2)) with erf from Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7, far below the
// bf16 rounding of the output): one v_rcp, one v_exp and a 5-term Horner chain instead of libm's branchy erff.
FP_DEVICE float gelu_erf(float x) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float e = __builtin_amdgcn_exp2f(-z * z * 1.44269504088896340736f);
  const float erf_abs = 1.f - p * t * e;
  return 0.5f * x * (1.f + copysignf(erf_abs, x));
}

// BM x BN block tile, WM x WN waves, each wave (BM/WM) x (BN/WN) = TM x TN MFMA tiles of 32x32.
// 16 rows x 32 bf16 (1 KiB) of a BK=32 sub-tile; swizzle (row>>2)&3 over the 4 chunks of a 64-B row.
FP_DEVICE void stage_rows32(const __bf16* __restrict__ g, int ld, int row0, int k0, char* lds, int rblk, int lane) {
  const int row = rblk * 16 + (lane >> 2);
  const int chunk = (lane & 3) ^ ((row >> 2) & 3);
  const __bf16* src = g + (size_t)(row0 + row) * ld + k0 + chunk * 8;
  __builtin_amdgcn_global_load_lds((gbl_cvoid*)src, (lds_void*)(lds + rblk * 1024), 16, 0, 0);
}
FP_DEVICE bf16x8 read_frag32(const char* lds, int row, int chunk) {
  return *reinterpret_cast<const bf16x8*>(lds + row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4));
}

template <int EPI, int BM, int BN, int WM, int WN, int PIPE>
__global__ __launch_bounds__(WM * WN * 64, 2) void gemm_bf16_kernel(GemmBf16Args a) {
  constexpr int NW = WM * WN, TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE = A_BYTES + B_BYTES;
  constexpr int A_INSTR = BM / 8 / NW, B_INSTR = BN / 8 / NW;  // DMA instructions per wave per K-tile
  static_assert(A_INSTR % 4 == 0 && B_INSTR % 4 == 0 || (A_INSTR + B_INSTR) % 4 == 0, "staging split");
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 stages][A | B]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN, l31 = lane & 31, kh = lane >> 5;

  const unsigned nwg = gridDim.x;
  const unsigned lid = xcd_remap(blockIdx.x, nwg);
  int m0, n0;
  if (a.tail_parent_tile == 0) {
    const unsigned tiles_n = a.N / BN, id = lid + a.tile_id_offset;
    m0 = (id / tiles_n) * BM;
    n0 = (id % tiles_n) * BN;
  } else {
    // tail launch: this grid covers the parent tiles [tile_id_offset, ...) of a (tail_parent_tile)^2 tiling,
    // each cut into (tail_parent_tile / BM) x (tail_parent_tile / BN) tiles of this kernel's size
    const unsigned pt = a.tail_parent_tile, sm = pt / BM, sn = pt / BN, per = sm * sn;
    const unsigned parent = a.tile_id_offset + lid / per, sub = lid % per;
    const unsigned ptiles_n = a.N / pt;
    m0 = (parent / ptiles_n) * pt + (sub / sn) * BM;
    n0 = (parent % ptiles_n) * pt + (sub % sn) * BN;
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  unsigned long long ts0 = 0, ts1 = 0, ts2 = 0;
  if (a.dbg) ts0 = __builtin_readcyclecounter();
  if constexpr (PIPE == 0) {
  // DMA piece q (0 .. A_INSTR+B_INSTR-1) of this wave for K-tile k0 into stage buffer `buf`
  auto stage_piece = [&](int q, int k0, char* buf) {
    if (q < A_INSTR) stage_rows(a.A, a.lda, m0, k0, buf, wave * A_INSTR + q, lane);
    else stage_rows(a.W, a.ldw, n0, k0, buf + A_BYTES, wave * B_INSTR + (q - A_INSTR), lane);
  };
  constexpr int PIECES = A_INSTR + B_INSTR, PER_KS = PIECES / 4;

  const int nk = a.K / BK;
#pragma unroll
  for (int q = 0; q < PIECES; ++q) stage_piece(q, 0, smem);

  for (int t = 0; t < nk; ++t) {
    const int cur = t & 1;
    __syncthreads();  // drains the DMA of tile t (vmcnt(0)) and fences the readers of the other stage
    if (a.dbg && t == 0) ts1 = __builtin_readcyclecounter();
    char* nxt = smem + (cur ^ 1) * STAGE;
    const bool more = t + 1 < nk;
    const char* As = smem + cur * STAGE;
    const char* Ws = As + A_BYTES;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      // next tile's DMA is issued in four slices, one ahead of each k-step's MFMAs
      if (more) {
#pragma unroll
        for (int q = 0; q < PER_KS; ++q) stage_piece(ks * PER_KS + q, (t + 1) * BK, nxt);
      }
      const int chunk = ks * 2 + kh;
      bf16x8 af[TM], wf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = read_frag(As, wm * (BM / WM) + i * 32 + l31, chunk);
#pragma unroll
      for (int j = 0; j < TN; ++j) wf[j] = read_frag(Ws, wn * (BN / WN) + j * 32 + l31, chunk);
      // swapped operands: D[i = n][j = m]
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
    }
  }

  } else if constexpr (PIPE == 7) {
    // ---- double-buffered BK=64, paired k-steps, with the next tile's DMA pieces interleaved BETWEEN the MFMAs of the
    // burst (one piece per two MFMAs): a global_load_lds costs its wave ~100-180 issue cycles, which overlap with the
    // matrix pipe only while that same wave has MFMAs executing (guide: "MFMA <-> buffer_load interleaved 1:1").
    constexpr int A_I = BM / 8 / NW, B_I = BN / 8 / NW, PCS = A_I + B_I;
    constexpr int STG = (BM + BN) * BK * 2;
    const int nk = a.K / BK;
    auto piece = [&](int q, int t) {
      char* buf = smem + (t & 1) * STG;
      if (q < A_I) stage_rows(a.A, a.lda, m0, t * BK, buf, wave * A_I + q, lane);
      else stage_rows(a.W, a.ldw, n0, t * BK, buf + BM * BK * 2, wave * B_I + (q - A_I), lane);
    };
#pragma unroll
    for (int q = 0; q < PCS; ++q) piece(q, 0);
    for (int t = 0; t < nk; ++t) {
      __syncthreads();
      const char* As = smem + (t & 1) * STG;
      const char* Ws = As + BM * BK * 2;
      const int tnext = t + 1 < nk ? t + 1 : t;
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        bf16x8 af[2][TM], wf[2][TN];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const int chunk = (half * 2 + s2) * 2 + kh;
#pragma unroll
          for (int i = 0; i < TM; ++i) af[s2][i] = read_frag(As, wm * (BM / WM) + i * 32 + l31, chunk);
#pragma unroll
          for (int j = 0; j < TN; ++j) wf[s2][j] = read_frag(Ws, wn * (BN / WN) + j * 32 + l31, chunk);
        }
        constexpr int NM = 2 * TM * TN;            // MFMAs of this burst
        constexpr int PER = PCS / 2;               // DMA pieces to place in this burst
        constexpr int EVERY = NM / PER;            // one piece after every EVERY MFMAs
#pragma unroll
        for (int idx = 0; idx < NM; ++idx) {
          const int s2 = idx / (TM * TN), i = (idx / TN) % TM, j = idx % TN;
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[s2][j], af[s2][i], acc[i][j], 0, 0, 0);
          // (the last tile re-stages itself into the idle buffer: branch-free bursts, 1/nk extra L2 traffic)
          if ((idx % EVERY) == EVERY - 1 && idx / EVERY < PER) piece(half * PER + idx / EVERY, tnext);
        }
#pragma unroll
        for (int g = 0; g < PER; ++g) {
          __builtin_amdgcn_sched_group_barrier(0x008, EVERY, 0);  // EVERY MFMAs
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);      // then one VMEM read (the LDS-DMA piece)
        }
      }
    }
  } else if constexpr (PIPE == 8) {
    // ---- as PIPE 7, plus the fragment reads of the second half of a K-tile are issued between the MFMAs of the
    // first half's burst (one ds_read_b128 per MFMA), so only one of the two LDS read phases per tile is exposed.
    constexpr int A_I = BM / 8 / NW, B_I = BN / 8 / NW, PCS = A_I + B_I;
    constexpr int STG = (BM + BN) * BK * 2;
    const int nk = a.K / BK;
    auto piece = [&](int q, int t) {
      char* buf = smem + (t & 1) * STG;
      if (q < A_I) stage_rows(a.A, a.lda, m0, t * BK, buf, wave * A_I + q, lane);
      else stage_rows(a.W, a.ldw, n0, t * BK, buf + BM * BK * 2, wave * B_I + (q - A_I), lane);
    };
#pragma unroll
    for (int q = 0; q < PCS; ++q) piece(q, 0);
    constexpr int NM = 2 * TM * TN, PER = PCS / 2, EVERY = NM / PER, NF = 2 * (TM + TN);
    for (int t = 0; t < nk; ++t) {
      __syncthreads();
      const char* As = smem + (t & 1) * STG;
      const char* Ws = As + BM * BK * 2;
      const int tnext = t + 1 < nk ? t + 1 : t;
      bf16x8 af[2][2][TM], wf[2][2][TN];  // [half][k-step]
      auto read_one = [&](int half, int f) {  // f-th fragment (0 .. NF-1) of a half
        const int s2 = f / (TM + TN), r = f % (TM + TN);
        const int chunk = (half * 2 + s2) * 2 + kh;
        if (r < TM) af[half][s2][r] = read_frag(As, wm * (BM / WM) + r * 32 + l31, chunk);
        else wf[half][s2][r - TM] = read_frag(Ws, wn * (BN / WN) + (r - TM) * 32 + l31, chunk);
      };
#pragma unroll
      for (int f = 0; f < NF; ++f) read_one(0, f);
#pragma unroll
      for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int idx = 0; idx < NM; ++idx) {
          const int s2 = idx / (TM * TN), i = (idx / TN) % TM, j = idx % TN;
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[half][s2][j], af[half][s2][i], acc[i][j], 0, 0, 0);
          if (half == 0 && idx < NF) read_one(1, idx);
          if ((idx % EVERY) == EVERY - 1 && idx / EVERY < PER) piece(half * PER + idx / EVERY, tnext);
        }
        if (half == 0) {
#pragma unroll
          for (int g = 0; g < NM; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                      // 1 MFMA
            if (g < NF) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);          // 1 DS read
            if ((g % EVERY) == EVERY - 1) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // 1 VMEM read (DMA piece)
          }
        } else {
#pragma unroll
          for (int g = 0; g < PER; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, EVERY, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
          }
        }
      }
    }
  } else if constexpr (PIPE == 6) {
    // ---- role-split schedule (needs WM == 2: wave w and w + NW/2 share a SIMD).  The two wave rows run one barrier
    // apart: in every barrier interval ("slot") one row issues its 16 MFMAs of a half K-tile while the other row
    // does its LDS fragment reads (+ the DMA issue) for the next half, then they swap.  The matrix pipe of each SIMD
    // always has exactly one wave feeding it and the LDS latency of the other wave is off the critical path
    // (guide section 5: 8-phase idea, here with 4 slots per K-tile and 32x32x16 MFMAs).
    //   row 0: slot 4t: L0(t)  4t+1: C0(t)  4t+2: L1(t)  4t+3: C1(t)
    //   row 1: slot 4t+1: L0(t) ...                                  4t+4: C1(t)
    //   DMA of tile u is issued by every wave in slot 4u-4 (stage u&1 was last read in slot 4u-5) and waited for
    //   (vmcnt(0)) right before the barrier that ends slot 4u-1.
    static_assert(WM == 2, "role-split schedule assumes two wave rows");
    constexpr int A_I = BM / 8 / NW, B_I = BN / 8 / NW;
    constexpr int STG = (BM + BN) * BK * 2;
    const int nk = a.K / BK;
    const int grp = wm;  // wave row = role group
    auto issue_tile = [&](int t) {
      char* buf = smem + (t & 1) * STG;
#pragma unroll
      for (int q = 0; q < A_I; ++q) stage_rows(a.A, a.lda, m0, t * BK, buf, wave * A_I + q, lane);
#pragma unroll
      for (int q = 0; q < B_I; ++q) stage_rows(a.W, a.ldw, n0, t * BK, buf + BM * BK * 2, wave * B_I + q, lane);
    };
    bf16x8 af[2][TM], wf[2][TN];
    auto load_half = [&](const char* As, int half) {
      const char* Ws = As + BM * BK * 2;
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const int chunk = (half * 2 + s2) * 2 + kh;
#pragma unroll
        for (int i = 0; i < TM; ++i) af[s2][i] = read_frag(As, wm * (BM / WM) + i * 32 + l31, chunk);
#pragma unroll
        for (int j = 0; j < TN; ++j) wf[s2][j] = read_frag(Ws, wn * (BN / WN) + j * 32 + l31, chunk);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // fragments in registers before the hand-over barrier
    };
    auto compute_half = [&]() {
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[s2][j], af[s2][i], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
    };
    // prologue: tile 0 by everyone; row 1 also issues tile 1 now (its regular slot would be "slot 0", which it idles)
    issue_tile(0);
    if (grp == 1 && nk > 1) {
      issue_tile(1);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(A_I + B_I) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();                 // slot 0 starts: tile 0 is in LDS
    if (grp == 1) __builtin_amdgcn_s_barrier();   // row 1 runs one slot behind
    for (int t = 0; t < nk; ++t) {
      const char* As = smem + (t & 1) * STG;
      // ---- L0(t)
      if (grp == 0 && t + 1 < nk) issue_tile(t + 1);          // slot 4t = 4(t+1)-4
      load_half(As, 0);
      __builtin_amdgcn_s_barrier();
      // ---- C0(t)
      compute_half();
      __builtin_amdgcn_s_barrier();
      // ---- L1(t)
      load_half(As, 1);
      if (grp == 1 && t + 1 < nk) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // row 1 is in slot 4t+3 = 4(t+1)-1
      __builtin_amdgcn_s_barrier();
      // ---- C1(t)
      if (grp == 1 && t + 2 < nk) issue_tile(t + 2);          // row 1's C1(t) is slot 4t+4 = 4(t+2)-4
      compute_half();
      if (grp == 0 && t + 1 < nk) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // row 0 is in slot 4t+3 = 4(t+1)-1
      __builtin_amdgcn_s_barrier();
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();   // balance row 1's extra barrier
  } else if constexpr (PIPE == 5) {
    // ---- double-buffered BK=64, k-steps fused in pairs: 12 fragment reads, then 16 MFMAs.  With two waves per SIMD
    // one wave's MFMA burst (16 x ~24 cycles when the pipe alternates waves) is long enough to cover the partner's
    // LDS read latency, so the two waves settle into anti-phase instead of both idling on lgkmcnt.
    constexpr int A_I = BM / 8 / NW, B_I = BN / 8 / NW, PCS = A_I + B_I;
    constexpr int STG = (BM + BN) * BK * 2;
    const int nk = a.K / BK;
    auto piece = [&](int q, int t) {
      char* buf = smem + (t & 1) * STG;
      if (q < A_I) stage_rows(a.A, a.lda, m0, t * BK, buf, wave * A_I + q, lane);
      else stage_rows(a.W, a.ldw, n0, t * BK, buf + BM * BK * 2, wave * B_I + (q - A_I), lane);
    };
#pragma unroll
    for (int q = 0; q < PCS; ++q) piece(q, 0);
    for (int t = 0; t < nk; ++t) {
      __syncthreads();
      const char* As = smem + (t & 1) * STG;
      const char* Ws = As + BM * BK * 2;
      const bool more = t + 1 < nk;
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        if (more) {
#pragma unroll
          for (int q = 0; q < PCS / 2; ++q) piece(half * (PCS / 2) + q, t + 1);
        }
        bf16x8 af[2][TM], wf[2][TN];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const int chunk = (half * 2 + s2) * 2 + kh;
#pragma unroll
          for (int i = 0; i < TM; ++i) af[s2][i] = read_frag(As, wm * (BM / WM) + i * 32 + l31, chunk);
#pragma unroll
          for (int j = 0; j < TN; ++j) wf[s2][j] = read_frag(Ws, wn * (BN / WN) + j * 32 + l31, chunk);
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[s2][j], af[s2][i], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
      }
    }
  } else if constexpr (PIPE == 2) {
    // ---- double-buffered BK=64 with the fragment reads software-pipelined one k-step ahead ACROSS the tile
    // boundary: the per-tile barrier (and the first LDS reads of the next tile) sit in front of the last
    // k-step's MFMAs of the current tile, so the barrier bubble is covered by matrix work.
    constexpr int A_I = BM / 8 / NW, B_I = BN / 8 / NW;
    constexpr int STG = (BM + BN) * BK * 2;
    const int nk = a.K / BK;
    auto issue_tile = [&](int t) {
      char* buf = smem + (t & 1) * STG;
#pragma unroll
      for (int q = 0; q < A_I; ++q) stage_rows(a.A, a.lda, m0, t * BK, buf, wave * A_I + q, lane);
#pragma unroll
      for (int q = 0; q < B_I; ++q) stage_rows(a.W, a.ldw, n0, t * BK, buf + BM * BK * 2, wave * B_I + q, lane);
    };
    bf16x8 af[2][TM], wf[2][TN];
    auto read_frags = [&](int slot, const char* As, int ks) {
      const char* Ws = As + BM * BK * 2;
      const int chunk = ks * 2 + kh;
#pragma unroll
      for (int i = 0; i < TM; ++i) af[slot][i] = read_frag(As, wm * (BM / WM) + i * 32 + l31, chunk);
#pragma unroll
      for (int j = 0; j < TN; ++j) wf[slot][j] = read_frag(Ws, wn * (BN / WN) + j * 32 + l31, chunk);
    };
    issue_tile(0);
    if (nk > 1) issue_tile(1);
    if (nk > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(A_I + B_I) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    read_frags(0, smem, 0);
    for (int t = 0; t < nk; ++t) {
      const char* As = smem + (t & 1) * STG;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int cs = ks & 1;
        if (ks < 3) {
          read_frags(cs ^ 1, As, ks + 1);
        } else if (t + 1 < nk) {
          __syncthreads();  // tile t+1 landed; every wave's reads of tile t have completed
          if (t + 2 < nk) issue_tile(t + 2);
          read_frags(cs ^ 1, smem + ((t + 1) & 1) * STG, 0);
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[cs][j], af[cs][i], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
      }
    }
  } else {
    constexpr int DEPTH = PIPE == 3 ? 2 : (PIPE == 4 ? 3 : 4), DIST = DEPTH - 1;
    // ---- DEPTH-deep ring of BK=32 sub-stages, DMA issued 3 sub-stages ahead, counted vmcnt + raw s_barrier:
    // HBM/L2 latency is covered by ~3 sub-stages of MFMA work instead of one K-tile (guide section 5 T3+T4).
    constexpr int SA = BM * 64, SB = BN * 64, SUB = SA + SB;          // bytes per sub-stage
    constexpr int AI = BM / 16 / NW, BI = BN / 16 / NW;               // DMA instructions per wave per sub-stage
    static_assert(AI >= 1 && BI >= 1, "tile too small for the wave count");
    const int ns = a.K / 32;
    auto issue = [&](int sidx) {
      char* buf = smem + (sidx % DEPTH) * SUB;
#pragma unroll
      for (int q = 0; q < AI; ++q) stage_rows32(a.A, a.lda, m0, sidx * 32, buf, wave * AI + q, lane);
#pragma unroll
      for (int q = 0; q < BI; ++q) stage_rows32(a.W, a.ldw, n0, sidx * 32, buf + SA, wave * BI + q, lane);
    };
#pragma unroll
    for (int i = 0; i < DIST; ++i)
      if (i < ns) issue(i);
    for (int sidx = 0; sidx < ns; ++sidx) {
      // wait until this wave's pieces of sub-stage `sidx` have landed (younger groups stay in flight)
      const int younger = min(DIST - 1, ns - 1 - sidx);
      if (DIST >= 3 && younger == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (AI + BI)) : "memory");
      else if (DIST >= 2 && younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AI + BI) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();  // everyone's pieces landed; everyone is done reading slot (sidx-1)&3
      if (sidx + DIST < ns) issue(sidx + DIST);
      const char* As = smem + (sidx % DEPTH) * SUB;
      const char* Ws = As + SA;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int chunk = ks * 2 + kh;
        bf16x8 af[TM], wf[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i] = read_frag32(As, wm * (BM / WM) + i * 32 + l31, chunk);
#pragma unroll
        for (int j = 0; j < TN; ++j) wf[j] = read_frag32(Ws, wn * (BN / WN) + j * 32 + l31, chunk);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
      }
    }
  }

  if (a.dbg) ts2 = __builtin_readcyclecounter();
  // ---- epilogue: acc[tm][tn][r] = C[m][n],  m = m0 + wm*(BM/WM) + tm*32 + (lane&31),
  //      n = n0 + wn*(BN/WN) + tn*32 + 8*(r>>2) + 4*(lane>>5) + (r&3)
  // A lane owns pieces of 32 different rows, so storing straight from registers makes every store instruction touch
  // 32 cache lines (the address path, not HBM, then bounds the tail).  Instead each 32-row band of the tile goes
  // through LDS (free after the main loop) and leaves as whole rows: 16 B per lane, lane-contiguous.
  // bf16 outputs leave through an LDS slab (whole-row 16-B stores: the tail drops from ~15k to ~8k cycles per tile);
  // the fp32 read-modify-write epilogues are bound by the residual traffic itself and stay register-direct.
  constexpr bool USE_SLAB = EPI == GEMM_EPI_BIAS_BF16 || EPI == GEMM_EPI_GELU_BF16 || EPI == GEMM_EPI_QKV_BF16 ||
                            EPI == GEMM_EPI_SWIGLU_BF16;
  if constexpr (USE_SLAB) {
  constexpr bool OUT_F32 = EPI == GEMM_EPI_LS_RESID_F32 || EPI == GEMM_EPI_TOKENS_F32 || EPI == GEMM_EPI_BIAS_F32;
  constexpr int ESZ = OUT_F32 ? 4 : 2;
  constexpr int OUT_COLS = EPI == GEMM_EPI_SWIGLU_BF16 ? BN / 2 : BN;  // SwiGLU folds column pairs
  constexpr int SLAB_ROWS = WM * 32, SLAB_STRIDE = OUT_COLS * ESZ + 16;  // +16 B: de-phases the rows across LDS banks
  constexpr int CHUNKS_PER_ROW = OUT_COLS * ESZ / 16, SLAB_CHUNKS = SLAB_ROWS * CHUNKS_PER_ROW, NT = NW * 64;
  static_assert(SLAB_ROWS * SLAB_STRIDE <= (BM + BN) * 64 * (PIPE == 3 ? 2 : (PIPE == 4 ? 3 : 4)), "slab must fit the main-loop LDS");
  float4 bias[TN][4], gam[TN][4];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int n = n0 + wn * (BN / WN) + tn * 32 + 8 * g + 4 * kh;
      bias[tn][g] = *reinterpret_cast<const float4*>(a.bias + n);
      if constexpr (EPI == GEMM_EPI_LS_RESID_F32) gam[tn][g] = *reinterpret_cast<const float4*>(a.gamma + n);
    }
  bool v_tile = false;  // qkv: tiles inside the V column block scatter V^T straight from registers
  if constexpr (EPI == GEMM_EPI_QKV_BF16) v_tile = n0 >= 2 * a.vit_dim;
  __syncthreads();  // every wave is done with the operand tiles in LDS
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    const int m = m0 + wm * (BM / WM) + tm * 32 + l31;
    if (v_tile) {
      if constexpr (EPI == GEMM_EPI_QKV_BF16) {
        if (m < a.M_valid) {
          const int vb = m / a.tok_n, vt = m - vb * a.tok_n;
#pragma unroll
          for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const int nn = n0 + wn * (BN / WN) + tn * 32 + 8 * g + 4 * kh - 2 * a.vit_dim;  // head*64 + d
              const float4 bs = bias[tn][g];
              __bf16* vtp = a.vt + ((size_t)vb * a.vit_dim + nn) * a.vt_ld + vt;
              vtp[0 * (size_t)a.vt_ld] = (__bf16)(acc[tm][tn][4 * g + 0] + bs.x);
              vtp[1 * (size_t)a.vt_ld] = (__bf16)(acc[tm][tn][4 * g + 1] + bs.y);
              vtp[2 * (size_t)a.vt_ld] = (__bf16)(acc[tm][tn][4 * g + 2] + bs.z);
              vtp[3 * (size_t)a.vt_ld] = (__bf16)(acc[tm][tn][4 * g + 3] + bs.w);
            }
        }
      }
      continue;
    }
    // (a) registers -> slab (final values except for the operand that needs a global read)
    char* srow = smem + (wm * 32 + l31) * SLAB_STRIDE;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int col = wn * (BN / WN) + tn * 32 + 8 * g + 4 * kh;
        const float4 bs = bias[tn][g];
        float v0 = acc[tm][tn][4 * g + 0] + bs.x, v1 = acc[tm][tn][4 * g + 1] + bs.y;
        float v2 = acc[tm][tn][4 * g + 2] + bs.z, v3 = acc[tm][tn][4 * g + 3] + bs.w;
        if constexpr (EPI == GEMM_EPI_GELU_BF16) {
          v0 = gelu_erf(v0); v1 = gelu_erf(v1); v2 = gelu_erf(v2); v3 = gelu_erf(v3);
        }
        if constexpr (EPI == GEMM_EPI_LS_RESID_F32) {
          const float4 gm = gam[tn][g];
          v0 *= gm.x; v1 *= gm.y; v2 *= gm.z; v3 *= gm.w;
        }
        if constexpr (EPI == GEMM_EPI_SWIGLU_BF16) {
          const float h0 = v0 / (1.f + __builtin_amdgcn_exp2f(-v0 * 1.44269504088896340736f)) * v1;  // silu(x1) * x2
          const float h1 = v2 / (1.f + __builtin_amdgcn_exp2f(-v2 * 1.44269504088896340736f)) * v3;
          *reinterpret_cast<unsigned*>(srow + (col >> 1) * 2) = pack_bf16x2(h0, h1);
        } else if constexpr (OUT_F32) *reinterpret_cast<float4*>(srow + col * 4) = make_float4(v0, v1, v2, v3);
        else *reinterpret_cast<uint2*>(srow + col * 2) = make_uint2(pack_bf16x2(v0, v1), pack_bf16x2(v2, v3));
      }
    __syncthreads();
    // (b) slab -> global, whole rows.  Global reads (residual / pos-embed rows) of all passes are issued first,
    // so the tail pays one memory round trip per band instead of one per pass.
    constexpr int PASSES = (SLAB_CHUNKS + NT - 1) / NT;
    static_assert(SLAB_CHUNKS % NT == 0, "slab chunks must divide evenly over the block");
    float4 ext[PASSES];
    size_t orow[PASSES];
    bool ok[PASSES];
#pragma unroll
    for (int it = 0; it < PASSES; ++it) {
      const int id = tid + it * NT;
      const int r = id / CHUNKS_PER_ROW, c = id - r * CHUNKS_PER_ROW;
      const int gm_row = m0 + (r >> 5) * (BM / WM) + tm * 32 + (r & 31);
      ok[it] = gm_row < a.M_valid;
      orow[it] = gm_row;
      if constexpr (EPI == GEMM_EPI_TOKENS_F32) {
        const int b = gm_row / a.tok_np, pidx = gm_row - b * a.tok_np;
        orow[it] = (size_t)b * a.tok_n + a.tok_skip + pidx;
        if (ok[it]) ext[it] = *reinterpret_cast<const float4*>(a.pos + (size_t)pidx * a.ldo + n0 + c * 4);
      }
      if constexpr (EPI == GEMM_EPI_LS_RESID_F32) {
        if (ok[it]) ext[it] = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(a.out) + orow[it] * a.ldo + n0 + c * 4);
      }
    }
#pragma unroll
    for (int it = 0; it < PASSES; ++it) {
      const int id = tid + it * NT;
      const int r = id / CHUNKS_PER_ROW, c = id - r * CHUNKS_PER_ROW;
      if (!ok[it]) continue;
      const char* sp = smem + r * SLAB_STRIDE + c * 16;
      if constexpr (!OUT_F32) {
        const int ncol = (EPI == GEMM_EPI_SWIGLU_BF16 ? n0 / 2 : n0) + c * 8;
        *reinterpret_cast<uint4*>(reinterpret_cast<__bf16*>(a.out) + orow[it] * a.ldo + ncol) = *reinterpret_cast<const uint4*>(sp);
      } else {
        float4 v = *reinterpret_cast<const float4*>(sp);
        if constexpr (EPI == GEMM_EPI_LS_RESID_F32 || EPI == GEMM_EPI_TOKENS_F32) {
          v.x += ext[it].x; v.y += ext[it].y; v.z += ext[it].z; v.w += ext[it].w;
        }
        *reinterpret_cast<float4*>(reinterpret_cast<float*>(a.out) + orow[it] * a.ldo + n0 + c * 4) = v;
      }
    }
    if (tm + 1 < TM) __syncthreads();
  }
  } else {
  float4 bias[TN][4], gam[TN][4];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int n = n0 + wn * (BN / WN) + tn * 32 + 8 * g + 4 * kh;
      bias[tn][g] = *reinterpret_cast<const float4*>(a.bias + n);
      if constexpr (EPI == GEMM_EPI_LS_RESID_F32) gam[tn][g] = *reinterpret_cast<const float4*>(a.gamma + n);
    }
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    const int m = m0 + wm * (BM / WM) + tm * 32 + l31;
    if (m >= a.M_valid) continue;
    size_t out_row = m;
    int vb = 0, vt = 0, pidx = 0;
    if constexpr (EPI == GEMM_EPI_TOKENS_F32) {
      const int b = m / a.tok_np;
      pidx = m - b * a.tok_np;
      out_row = (size_t)b * a.tok_n + a.tok_skip + pidx;
    }
    if constexpr (EPI == GEMM_EPI_QKV_BF16) {
      vb = m / a.tok_n;
      vt = m - vb * a.tok_n;
    }
    float4 extra[TN][4];  // residual row (LS_RESID) or pos-embed row (TOKENS)
    if constexpr (EPI == GEMM_EPI_LS_RESID_F32 || EPI == GEMM_EPI_TOKENS_F32) {
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = n0 + wn * (BN / WN) + tn * 32 + 8 * g + 4 * kh;
          if constexpr (EPI == GEMM_EPI_LS_RESID_F32)
            extra[tn][g] = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(a.out) + out_row * a.ldo + n);
          else
            extra[tn][g] = *reinterpret_cast<const float4*>(a.pos + (size_t)pidx * a.ldo + n);
        }
    }
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = n0 + wn * (BN / WN) + tn * 32 + 8 * g + 4 * kh;
        const float4 bs = bias[tn][g];
        float v0 = acc[tm][tn][4 * g + 0] + bs.x, v1 = acc[tm][tn][4 * g + 1] + bs.y;
        float v2 = acc[tm][tn][4 * g + 2] + bs.z, v3 = acc[tm][tn][4 * g + 3] + bs.w;
        if constexpr (EPI == GEMM_EPI_BIAS_BF16 || EPI == GEMM_EPI_GELU_BF16 || EPI == GEMM_EPI_QKV_BF16) {
          if constexpr (EPI == GEMM_EPI_GELU_BF16) {
            v0 = gelu_erf(v0); v1 = gelu_erf(v1); v2 = gelu_erf(v2); v3 = gelu_erf(v3);
          }
          bool transposed_v = false;
          if constexpr (EPI == GEMM_EPI_QKV_BF16) transposed_v = n >= 2 * a.vit_dim;
          if (!transposed_v) {
            uint2 pk = make_uint2(pack_bf16x2(v0, v1), pack_bf16x2(v2, v3));
            *reinterpret_cast<uint2*>(reinterpret_cast<__bf16*>(a.out) + out_row * a.ldo + n) = pk;
          } else {
            // V goes out transposed, Vt[b][head][d][t] (keys contiguous), for the attention P*V operand
            const int nn = n - 2 * a.vit_dim;  // head*64 + d
            __bf16* vtp = a.vt + ((size_t)vb * a.vit_dim + nn) * a.vt_ld + vt;
            vtp[0 * (size_t)a.vt_ld] = (__bf16)v0;
            vtp[1 * (size_t)a.vt_ld] = (__bf16)v1;
            vtp[2 * (size_t)a.vt_ld] = (__bf16)v2;
            vtp[3 * (size_t)a.vt_ld] = (__bf16)v3;
          }
        } else if constexpr (EPI == GEMM_EPI_LS_RESID_F32) {
          const float4 gm = gam[tn][g];
          float4 x = extra[tn][g];
          x.x += gm.x * v0; x.y += gm.y * v1; x.z += gm.z * v2; x.w += gm.w * v3;
          *reinterpret_cast<float4*>(reinterpret_cast<float*>(a.out) + out_row * a.ldo + n) = x;
        } else if constexpr (EPI == GEMM_EPI_TOKENS_F32) {
          const float4 pe = extra[tn][g];
          *reinterpret_cast<float4*>(reinterpret_cast<float*>(a.out) + out_row * a.ldo + n) =
              make_float4(v0 + pe.x, v1 + pe.y, v2 + pe.z, v3 + pe.w);
        } else if constexpr (EPI == GEMM_EPI_BIAS_F32) {
          *reinterpret_cast<float4*>(reinterpret_cast<float*>(a.out) + out_row * a.ldo + n) = make_float4(v0, v1, v2, v3);
        }
      }
  }
  }
  if (a.dbg) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long ts3 = __builtin_readcyclecounter();
    if (tid == 0) {
      unsigned long long* d = a.dbg + (size_t)blockIdx.x * 4;
      d[0] = ts0; d[1] = ts1; d[2] = ts2; d[3] = ts3;
    }
  }
}

template <int EPI, int BM, int BN, int WM, int WN, int PIPE>
int launch_cfg(const GemmBf16Args& a, hipStream_t st, unsigned grid_override = 0) {
  const unsigned grid = grid_override ? grid_override : (a.M / BM) * (a.N / BN);
  // 2 x BK=64 stages == 4 x BK=32 sub-stages; PIPE 3 / 4: ring of 2 / 3 sub-stages
  const size_t lds = (size_t)(BM + BN) * 64 * (PIPE == 3 ? 2 : (PIPE == 4 ? 3 : 4));
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_kernel<EPI, BM, BN, WM, WN, PIPE>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr = true;
  }
  hipLaunchKernelGGL((gemm_bf16_kernel<EPI, BM, BN, WM, WN, PIPE>), dim3(grid), dim3(WM * WN * 64), lds, st, a);
  FP_CHECK_LAUNCH("gemm_bf16_kernel");
  return FP_OK;
}

// Tile selection: 256x256 (8 waves, 1 block/CU, 128 KiB LDS) when the shape allows it and fills the chip,
// otherwise 128x128 (4 waves, 2 blocks/CU).
template <int EPI>
int launch(const GemmBf16Args& a, hipStream_t st) {
  const int force = a.tile_override;
  const bool big_ok = a.M % 256 == 0 && a.N % 256 == 0;
  const bool use_big = force == 256 || (force == 0 && big_ok && (a.M / 256) * (a.N / 256) >= 256);
  // pipe_override: 0 / 1 = default: plain double buffer (DMA of tile t+1 issued in 4 slices between the k-steps).
  // Alternatives kept for A/B: 2 = ring of BK=32 sub-stages with counted vmcnt, 3 = software-pipelined fragment reads,
  // 4 = paired k-steps, 5 = role-split wave rows, 6 = DMA pieces interleaved between the MFMAs, 7 = 6 + fragment reads
  // interleaved.  In isolation 6 is 5-8 % faster on random operands; inside the ViT pipeline all of them land within
  // 1.5 % of each other (whole-pipeline A/B on one box: 899 / 888 / 885 detections/s for 1 / 6 / 4).
  // FP_GEMM_PIPE (read once) overrides the default main loop for whole-pipeline A/B runs
  static const int env_pipe = getenv("FP_GEMM_PIPE") ? atoi(getenv("FP_GEMM_PIPE")) : 0;
  const int po = a.pipe_override ? a.pipe_override : env_pipe;
  const int pv = po == 2 ? 1 : (po == 3 ? 2 : (po == 4 ? 5 : (po == 5 ? 6 : (po == 6 ? 7 : (po == 7 ? 8 : 0)))));
  if (use_big && big_ok && pv == 5 && a.tail_split) {  // measured 2-5 % SLOWER than one launch on the ViT-L shapes: off by default
    // Tail balancing: 256^2 tiles for whole rounds of 256 CUs, the leftover parent tiles as 128^2 tiles at two
    // workgroups per CU (a partial last round of big tiles otherwise idles up to 255 CUs for a full tile time).
    const unsigned tiles = (a.M / 256) * (a.N / 256), full = tiles / 256 * 256, rest = tiles - full;
    if (full > 0 && rest > 0) {
      int rc = launch_cfg<EPI, 256, 256, 2, 4, 5>(a, st, full);
      if (rc != FP_OK) return rc;
      GemmBf16Args t = a;
      t.tile_id_offset = full;
      t.tail_parent_tile = 256;
      return launch_cfg<EPI, 128, 128, 2, 2, 5>(t, st, rest * 4);
    }
  }
  if (use_big && big_ok) {
    if (pv == 0) return launch_cfg<EPI, 256, 256, 2, 4, 0>(a, st);
    if (pv == 1) return launch_cfg<EPI, 256, 256, 2, 4, 1>(a, st);
    if (pv == 5) return launch_cfg<EPI, 256, 256, 2, 4, 5>(a, st);
    if (pv == 6) return launch_cfg<EPI, 256, 256, 2, 4, 6>(a, st);
    if (pv == 7) return launch_cfg<EPI, 256, 256, 2, 4, 7>(a, st);
    if (pv == 8) return launch_cfg<EPI, 256, 256, 2, 4, 8>(a, st);
    return launch_cfg<EPI, 256, 256, 2, 4, 2>(a, st);
  }
  if (force == 384 && a.N % 256 == 0) return launch_cfg<EPI, 128, 256, 2, 2, 4>(a, st);  // 4 waves, 72 KiB ring: 2 workgroups per CU
  if (pv == 0) return launch_cfg<EPI, 128, 128, 2, 2, 0>(a, st);
  if (pv == 1) return launch_cfg<EPI, 128, 128, 2, 2, 1>(a, st);
  if (pv == 7 || pv == 8) return launch_cfg<EPI, 128, 128, 2, 2, 7>(a, st);
  if (pv == 5 || pv == 6) return launch_cfg<EPI, 128, 128, 2, 2, 5>(a, st);
  return launch_cfg<EPI, 128, 128, 2, 2, 2>(a, st);
}

}  // namespace

int gemm_bf16_launch(int epi, const GemmBf16Args& a, hipStream_t st) {
  FP_REQUIRE(a.M > 0 && a.M % 128 == 0, "gemm_bf16: M (%d) must be a positive multiple of 128 (pad the activation buffer)", a.M);
  FP_REQUIRE(a.N > 0 && a.N % 128 == 0, "gemm_bf16: N (%d) must be a multiple of 128", a.N);
  FP_REQUIRE(a.K > 0 && a.K % BK == 0, "gemm_bf16: K (%d) must be a multiple of %d", a.K, BK);
  FP_REQUIRE(a.bias != nullptr, "gemm_bf16: bias is required (pass zeros)");
  FP_REQUIRE(a.lda % 8 == 0 && a.ldw % 8 == 0 && a.ldo % 4 == 0, "gemm_bf16: leading dims must keep 16-byte alignment");
  switch (epi) {
    case GEMM_EPI_BIAS_BF16: return launch<GEMM_EPI_BIAS_BF16>(a, st);
    case GEMM_EPI_GELU_BF16: return launch<GEMM_EPI_GELU_BF16>(a, st);
    case GEMM_EPI_QKV_BF16: return launch<GEMM_EPI_QKV_BF16>(a, st);
    case GEMM_EPI_LS_RESID_F32: return launch<GEMM_EPI_LS_RESID_F32>(a, st);
    case GEMM_EPI_TOKENS_F32: return launch<GEMM_EPI_TOKENS_F32>(a, st);
    case GEMM_EPI_BIAS_F32: return launch<GEMM_EPI_BIAS_F32>(a, st);
    case GEMM_EPI_SWIGLU_BF16: return launch<GEMM_EPI_SWIGLU_BF16>(a, st);
  }
  fp_set_error("gemm_bf16: unknown epilogue %d", epi);
  return FP_ERR_INVALID;
}
