// bf16 MFMA GEMM for the ViT linear layers: C = A[M,K] * W[N,K]^T (+ fused epilogue), fp32 accumulate.
//
// This is the arithmetic the reference delegates to the DINOv2 backbone's nn.Linear layers
// (call site /root/reference/utils/dinov2_utils.py:257).  MI355X design:
//   * block tile 256x256x64, 512 threads = 2x4 waves, each wave 128x64 = 4x2 v_mfma_f32_32x32x16_bf16
//     (128x128, 4 waves for small shapes)
//   * A and W tiles go HBM/L2 -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds, 16 B/lane, no VGPR round trip), double
//     buffered, one barrier per K-tile, next tile's DMA in flight under the MFMAs; on the 8-wave tile only ONE wave
//     row issues the DMA (each SIMD hosts a wave of either row: the partner keeps the matrix pipe fed while the
//     issuing wave is blocked in its ~100-cycle DMA issues)
//   * LDS image is row-major [row][64 bf16]; bank conflicts of the ds_read_b128 fragment reads are
//     removed by XOR-swizzling the 16-B chunk index with (row>>1)&7 -- applied on the *source* address
//     (the DMA destination is lane-linear) and again on the read (guide section 5.4 rule 21)
//   * operands are fed to the MFMA swapped (W as the "A" operand) so each lane ends up with 4 consecutive
//     output columns of one row; every epilogue but the small fp32 ones leaves through an LDS slab as whole rows
//   * logical workgroup ids are remapped so that each XCD's L2 sees a contiguous run of tiles sharing A panels; wide
//     outputs use an 8 x 4 super-tile raster per XCD that keeps a group of W panels resident in its L2.
// What bounds it, and what was tried and did not help: DESIGN.md section 5 "GEMM analysis".
#include <cstdlib>

#include <cstring>
#include "common.hpp"
#include "kernels.hpp"

namespace {

constexpr int BK = 64;

typedef __attribute__((address_space(3))) void lds_void;

FP_DEVICE int swz(int row) { return (row >> 1) & 7; }

// One DMA instruction: 8 rows x 64 bf16 (1 KiB) of a tile, row group `rblk`, into the lane-linear LDS image.
// Buffer addressing: the matrix is a raw buffer resource (4 SGPRs), the lane supplies ONE dword -- its byte offset inside
// an 8-row group, constant for the whole kernel (two variants: the swizzle depends on the parity of the row group) --
// and everything that moves (tile origin, row group, K-tile) is a scalar offset.  Compared with global_load_lds on
// 64-bit per-lane addresses this halves the address data a wave pushes to the texture-address unit per instruction
// and removes the per-piece 64-bit VALU address arithmetic from the main loop.
FP_DEVICE unsigned stage_lane_offset(int ld, int lane, int parity) {
  const int row_l = lane >> 3;                                   // row inside the 8-row group
  const int chunk = (lane & 7) ^ ((4 * parity + (row_l >> 1)) & 7);  // = (lane & 7) ^ swz(rblk * 8 + row_l)
  return (unsigned)(row_l * ld + chunk * 8) * 2u;
}
FP_DEVICE void stage_rows(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, int ld, int row0, int k0, char* lds, int rblk) {
  const unsigned soff = (unsigned)((row0 + rblk * 8) * ld + k0) * 2u;  // uniform
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)(lds + rblk * 1024), 16, voff, soff, 0, 0);
}

FP_DEVICE bf16x8 read_frag(const char* lds, int row, int chunk) {
  return *reinterpret_cast<const bf16x8*>(lds + row * 128 + ((chunk ^ swz(row)) << 4));
}

// GELU(x) = x Phi(x) for the bf16 path, two elements per instruction (v_pk_fma_f32 / v_pk_mul_f32).
// Phi(x) - 0.5 = 0.5 erf(x / sqrt 2) is an odd degree-13 minimax polynomial on |x| <= 3.9 (input clamped there, where
// Phi is within 4.8e-5 of its limit): max |dPhi| = 8.3e-5, i.e. a relative error <= 1.7e-4 for x >= 0 -- an order of
// magnitude below the bf16 half-ulp (2e-3) of the stored result -- at 6 VALU instructions per element instead of the
// ~27 instruction-equivalents of an erf built from v_rcp + v_exp.  With 128 outputs per lane this epilogue was ~10 us
// of VALU time per 256x256 tile, a third of fc1's run time (exact-erf GELU stays in the fp32 path, f32_tile.hip).
FP_DEVICE f32x2 gelu_pk(f32x2 x) {
  constexpr float X = 3.9f;
  f32x2 xc;
  xc[0] = __builtin_amdgcn_fmed3f(x[0], -X, X);
  xc[1] = __builtin_amdgcn_fmed3f(x[1], -X, X);
  const f32x2 s = xc * xc;
  f32x2 q = f32x2{3.214934915e-08f, 3.214934915e-08f};
  q = __builtin_elementwise_fma(q, s, f32x2{-2.075321994e-06f, -2.075321994e-06f});
  q = __builtin_elementwise_fma(q, s, f32x2{5.740229389e-05f, 5.740229389e-05f});
  q = __builtin_elementwise_fma(q, s, f32x2{-9.056365499e-04f, -9.056365499e-04f});
  q = __builtin_elementwise_fma(q, s, f32x2{9.218766509e-03f, 9.218766509e-03f});
  q = __builtin_elementwise_fma(q, s, f32x2{-6.556460516e-02f, -6.556460516e-02f});
  q = __builtin_elementwise_fma(q, s, f32x2{3.986083959e-01f, 3.986083959e-01f});
  const f32x2 phi = __builtin_elementwise_fma(xc, q, f32x2{0.5f, 0.5f});
  return x * phi;
}

// ... and for the fp16 epilogue ("f16" mode; its stored result carries 11 bits): the same form with nine coefficients on |x| <= 4.4, fitted under the
// constraint that x q(x^2) reaches 0.5 at the clamp (Phi = 1 exactly above it, -2.6e-8 below: no tail error that grows with |x|) and weighted by |x| (what
// is minimised is the error of GELU itself; constrained Lawson iteration, last coefficient adjusted in fp32): max |gelu error| 3.8e-5 in fp32 evaluation
// against 4.0e-4 of the seven-coefficient form above -- below the rounding of an fp16 result of magnitude >= 0.08 -- at 7 VALU instructions per element
// against ~12 of the erf form below.  Same index agreement with the fp32 mode as the erf form in same-box runs (154 / 143 against 156 / 141 slots of 160).
FP_DEVICE f32x2 gelu_pk9(f32x2 x) {
  constexpr float X = 4.4f;
  f32x2 xc;
  xc[0] = __builtin_amdgcn_fmed3f(x[0], -X, X);
  xc[1] = __builtin_amdgcn_fmed3f(x[1], -X, X);
  const f32x2 s = xc * xc;
  f32x2 q = f32x2{3.569422188e-11f, 3.569422188e-11f};
  q = __builtin_elementwise_fma(q, s, f32x2{-3.754043298e-09f, -3.754043298e-09f});
  q = __builtin_elementwise_fma(q, s, f32x2{1.740845335e-07f, 1.740845335e-07f});
  q = __builtin_elementwise_fma(q, s, f32x2{-4.724864539e-06f, -4.724864539e-06f});
  q = __builtin_elementwise_fma(q, s, f32x2{8.429298032e-05f, 8.429298032e-05f});
  q = __builtin_elementwise_fma(q, s, f32x2{-1.054992317e-03f, -1.054992317e-03f});
  q = __builtin_elementwise_fma(q, s, f32x2{9.643027559e-03f, 9.643027559e-03f});
  q = __builtin_elementwise_fma(q, s, f32x2{-6.607642770e-02f, -6.607642770e-02f});
  q = __builtin_elementwise_fma(q, s, f32x2{3.987614810e-01f, 3.987614810e-01f});
  const f32x2 phi = __builtin_elementwise_fma(xc, q, f32x2{0.5f, 0.5f});
  return x * phi;
}

// GELU in its erf form at fp32 accuracy (the f16x3 mode's fc1 epilogue; the reference's nn.GELU() inside the backbone's Mlp).
// erfc(z) exp(z^2) is a degree-7 polynomial in t = 1 / (1 + 0.3275911 z) on z >= 0 (the Abramowitz-Stegun 7.1.26 form with two more
// terms, refitted minimax against scipy's erfcx: |d erf| <= 3.5e-9 before rounding), and gelu(x) = x/2 + |x|/2 erf(|x| / sqrt 2), so
// no sign handling and no cancellation for x > 0.  Evaluated in fp32 on two elements per instruction (v_pk_fma_f32) + one v_rcp_f32 and
// one v_exp_f32 each: max |error| 4.2e-7 over [-12, 12] against the exact function -- closer to it than torch's own CPU gelu (1.2e-6,
// tools/gelu_accuracy.py) and than the f16x3 products (2^-22 relative) -- at ~12 instruction-equivalents per element; ocml's erff cost ~35
// and a sixth of the fc1 launch.
FP_DEVICE f32x2 gelu_erf_pk(f32x2 x) {
  const f32x2 hx = x * f32x2{0.5f, 0.5f};
  const f32x2 hax = {__builtin_fabsf(hx[0]), __builtin_fabsf(hx[1])};
  const f32x2 z = hax * f32x2{1.41421356237309504880f, 1.41421356237309504880f};   // |x| / sqrt 2
  const f32x2 den = __builtin_elementwise_fma(z, f32x2{0.3275911f, 0.3275911f}, f32x2{1.f, 1.f});
  const f32x2 t = {__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};
  f32x2 q = f32x2{-0.29844723923966754f, -0.29844723923966754f};
  q = __builtin_elementwise_fma(q, t, f32x2{1.5048025775340452f, 1.5048025775340452f});
  q = __builtin_elementwise_fma(q, t, f32x2{-2.0855161249300567f, -2.0855161249300567f});
  q = __builtin_elementwise_fma(q, t, f32x2{2.040068178110667f, 2.040068178110667f});
  q = __builtin_elementwise_fma(q, t, f32x2{-0.7490454190655901f, -0.7490454190655901f});
  q = __builtin_elementwise_fma(q, t, f32x2{0.4310967998728252f, 0.4310967998728252f});
  q = __builtin_elementwise_fma(q, t, f32x2{0.15704123123125485f, 0.15704123123125485f});
  q = q * t;
  const f32x2 a = z * (z * f32x2{-1.44269504088896340736f, -1.44269504088896340736f});
  const f32x2 e = {__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])};                 // exp(-z^2)
  const f32x2 erfz = __builtin_elementwise_fma(-q, e, f32x2{1.f, 1.f});                           // erf(|x| / sqrt 2)
  return __builtin_elementwise_fma(hax, erfz, hx);
}

// Sum over aligned groups of 32 lanes with DPP moves; the total is valid in the LAST lane of each group (lane & 31 == 31).
FP_DEVICE float row32_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xF, 0xF, true));  // row_ror:4
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, true));  // row_ror:8 -> every lane: its row of 16
  // row_bcast15 into rows 1 and 3 (row_mask 0xA): lane 15 of the row before -> lanes 16..31 / 48..63 hold the 32-lane total
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xA, 0xF, true));
  return v;
}

// ... and over aligned groups of 16 lanes (valid in every lane of the group)
FP_DEVICE float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xF, 0xF, true));  // row_ror:4
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, true));  // row_ror:8
  return v;
}

// BM x BN block tile, WM x WN waves, each wave (BM/WM) x (BN/WN) = TM x TN MFMA tiles of 32x32.
// F8: the operands are OCP fp8 (e4m3) instead of bf16.  An fp8 row of K elements is addressed as a bf16 row of K/2
// elements (the host passes K/2, lda/2, ldw/2), so a K-tile is the same 128-B-per-row LDS image holding 128 k-values and
// the staging code is shared; the tile is consumed by two v_mfma_scale_f32_32x32x64_f8f6f4 per accumulator (unit block
// scales: plain fp8 products, fp32 accumulation, twice the bf16 MFMA rate) whose operand -- lane (row, kh) holds k =
// 32 kh .. 32 kh + 31 of a 64-wide step, tools/ubench/fp8_probe.hip -- is two adjacent 16-B chunks of the row.
// Dequantisation lives in the epilogue: out = (acc + bias) * gamma with gamma = activation scale x per-channel weight
// scale (x LayerScale) and bias pre-divided by that scale on the host.
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) int i32x4;

// F8OUT (GELU / SwiGLU epilogues of the fp8 kernels): the result is itself the input of the next fp8 GEMM and leaves as
// e4m3(clamp(value * a.out_scale, +-448)) bytes, [M, N] (or [M, N/2]) with ldo in bytes -- no separate quantisation pass.
// SP (f16x3 mode): the operands are split-fp16 rows (common.hpp): a row of K logical elements is 2K halves, a K-tile is the same
// 128-B-per-row LDS image holding 32 k-values as [hi 32 | lo 32], the staging code is shared, and a K-tile is consumed by three
// v_mfma_f32_32x32x16_f16 per accumulator and 16-wide k-step -- hi*hi, hi*lo, lo*hi -- i.e. 3x the MFMAs and 2x the operand
// bytes of the bf16 kernel for products that carry 22 mantissa bits.  SPOUT: the BIAS / GELU / SwiGLU result is the next
// GEMM's (or the attention's) operand and leaves as a split-fp16 row scaled by a.out_scale; GELU is the exact erf form here.
// SX (f16f8 mode, with SP): the operands are f16f8 rows (common.hpp): per 64 logical k a 128-B tile of fp16 high halves (four 16-wide hi*hi steps)
// and a 128-B tile [e4m3(hi 2^-7) x 64 | e4m3(lo 2^4) x 64] consumed by two 64-wide fp8 MFMAs (lo_w * hi_a, hi_w * lo_a; block scale 2^3 on one
// operand) -- 8 instead of 12 fp16-MFMA units per 64 k at the same operand bytes.  The GELU / SwiGLU outputs (the next GEMM's A operand) leave as
// f16f8 rows; the BIAS output (q | k | v for the attention kernel) stays a split-fp16 row.
// H16 ("f16" mode): the bf16 kernel on IEEE fp16 operands -- v_mfma_f32_32x32x16_f16, fp16 outputs of the 16-bit epilogues and of the (hi, lo) residual stream
// (common.hpp pack_h2 / unpack_h2), GELU by the nine-coefficient polynomial gelu_pk9 (3.8e-5 absolute; the seven-coefficient one of the bf16 epilogue, 4e-4, sits
// below a bf16 half-ulp, not below an fp16 one).  An fp16 output beyond +-65504 becomes inf, and an inf poisons everything behind it (the row's residual stream, then --
// through the keys and values -- every token of the image): the pipeline's LAST kernel (final norm / sampling) reports non-finite features, nothing is tracked here.
// NSTAGE (the 64 x 128 tile of a one-crop batch only): K-tiles in flight per workgroup -- four instead of two, three tiles in flight behind the one being multiplied.
// Worth 4 % on the launch it was built for (fc2 at B = 1: 45.3 -> 43.5 us, tools/b1_gemm_probe.py): that launch is NOT latency-bound, as assumed -- 172 workgroups
// re-stream the 8 MB weight matrix 22 times (one pass per 64-row tile) at the ~35 GB/s one CU's LDS-DMA path sustains, 0.7 us per K-tile whatever the depth.
// The MFMA order per accumulator is untouched: the same bits.
template <int EPI, int BM, int BN, int WM, int WN, bool F8 = false, bool F8OUT = false, bool SP = false, bool SPOUT = false, bool SX = false, bool H16 = false, int NSTAGE = 2>
__global__ __launch_bounds__(WM * WN * 64, 2) void gemm_bf16_kernel(GemmBf16Args a) {
  static_assert(!SX || SP, "f16f8 rows are a form of the split operands");
  static_assert(NSTAGE == 2 || (NSTAGE == 4 && !F8 && !SP && WM * WN != 8), "the deep pipeline exists for the small bf16 / fp16 tiles");
  static_assert(!H16 || (!F8 && !SP), "plain fp16 operands exclude the fp8 and the split forms");
  static_assert(!F8OUT || (F8 && (EPI == GEMM_EPI_GELU_BF16 || EPI == GEMM_EPI_SWIGLU_BF16)), "fp8 output: GELU / SwiGLU epilogues of the fp8 kernels");
  static_assert(!(SP && F8) && (!SPOUT || SP), "split-fp16 and fp8 operands exclude each other; a split output needs split operands");
  static_assert(SPOUT == (SP && (EPI == GEMM_EPI_BIAS_BF16 || EPI == GEMM_EPI_GELU_BF16 || EPI == GEMM_EPI_SWIGLU_BF16)), "f16x3: the half-precision epilogues write split rows");
  constexpr int NW = WM * WN, NT = NW * 64, TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr bool HILO = EPI == GEMM_EPI_RESID_HILO;  // the residual stream as (hi, lo) bf16 arrays: hi IS the next GEMM's A operand
  constexpr bool RESID = EPI == GEMM_EPI_LS_RESID_F32 || EPI == GEMM_EPI_RESID_F32 || HILO;  // residual read-modify-write epilogues
  constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE = A_BYTES + B_BYTES;
  // DMA issue is asymmetric on the 8-wave tile: only the waves of row wm == 0 fetch (every SIMD hosts one wave of each
  // row).  A global/buffer_load..lds blocks its wave for ~60-180 issue cycles; when all eight waves issue their pieces
  // in lock step the matrix pipes idle meanwhile, when one wave per SIMD does it the other keeps its SIMD's pipe fed.
#ifdef FP_GEMM_ALT  // measurement build: every wave issues half as many pieces, the two wave rows in alternating halves of the K-tile
  constexpr bool ASYM = false;
  constexpr bool ALT = NW == 8;
#else
  constexpr bool ASYM = NW == 8;
  constexpr bool ALT = false;
#endif
  constexpr int NISSUE = ASYM ? NW / 2 : NW;
  constexpr int A_INSTR = BM / 8 / NISSUE, B_INSTR = BN / 8 / NISSUE;  // DMA instructions per issuing wave per K-tile
  constexpr int PIECES = A_INSTR + B_INSTR;
  static_assert(PIECES % 2 == 0, "staging split");  // quarters of the piece list per k-step (uneven for the 320-row tile: 18 pieces), halves in the fp8 / f16x3 loops
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 stages][A | B]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN, l31 = lane & 31, kh = lane >> 5;

  const unsigned nwg = gridDim.x;
  const unsigned lid = xcd_remap(blockIdx.x, nwg);
  const unsigned tiles_n = a.N / BN;
  int m0, n0;
  if (a.rast_r) {
    // super-tile raster: consecutive lids fill a (rast_r x rast_gn) super-tile, super-tiles run down M inside one
    // group of rast_gn n-tiles before moving to the next group (holes of the ragged last super-row exit at once)
    const unsigned per = a.rast_r * a.rast_gn, st = lid / per, r = lid - st * per;
    const unsigned SM = (a.m_tiles + a.rast_r - 1) / a.rast_r;
    const unsigned sn = st / SM, sm = st - sn * SM;
    const unsigned tm = sm * a.rast_r + r % a.rast_r, tn = sn * a.rast_gn + r / a.rast_r;
    if (tm >= (unsigned)a.m_tiles) return;
    m0 = tm * BM;
    n0 = tn * BN;
  } else {
    m0 = (lid / tiles_n) * BM;
    n0 = (lid % tiles_n) * BN;
  }
  const int kb = 0, ke = a.K / BK;

#ifdef FP_GEMM_TIMELINE  // tools/build_variant.sh -DFP_GEMM_TIMELINE: per-workgroup shader-clock stamps (never in the shipped library)
  unsigned long long ts0 = 0, ts1 = 0, ts2 = 0;
  if (a.dbg) ts0 = __builtin_readcyclecounter();
#endif
  {
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // DMA piece q (0 .. PIECES-1) of this wave for K-tile t into stage buffer `buf`
    static_assert(A_INSTR % 2 == 0 && B_INSTR % 2 == 0, "row-group parity of a piece must be a compile-time property");
    // the resources start at this tile's first row (32-bit offsets then never exceed one tile's extent, whatever the
    // size of the matrix) and end at the end of the matrix, clipped to the 4-GiB range of a buffer resource
    auto tile_rsrc = [](const __bf16* base, int row0, int rows, int ld) {
      const unsigned long long bytes = (unsigned long long)(rows - row0) * ld * 2ull;
      return __builtin_amdgcn_make_buffer_rsrc((void*)(base + (size_t)row0 * ld), 0, bytes > 0xffffffffull ? 0xffffffffu : (unsigned)bytes, 0x00020000);
    };
    const __amdgpu_buffer_rsrc_t rsrc_a = tile_rsrc(a.A, m0, a.M, a.lda), rsrc_w = tile_rsrc(a.W, n0, a.N, a.ldw);
    const unsigned voff_a[2] = {stage_lane_offset(a.lda, lane, 0), stage_lane_offset(a.lda, lane, 1)};
    const unsigned voff_w[2] = {stage_lane_offset(a.ldw, lane, 0), stage_lane_offset(a.ldw, lane, 1)};
    const bool issuer = !ASYM || wm == 0;  // wave-uniform
    const int iw = ASYM ? wave % NISSUE : wave;
    auto stage_piece = [&](int q, int t, char* buf) {
      if (!issuer) return;
      if (q < A_INSTR) stage_rows(rsrc_a, voff_a[q & 1], a.lda, 0, t * BK, buf, iw * A_INSTR + q);
      else stage_rows(rsrc_w, voff_w[(q - A_INSTR) & 1], a.ldw, 0, t * BK, buf + A_BYTES, iw * B_INSTR + (q - A_INSTR));
    };
#pragma unroll
    for (int q = 0; q < PIECES; ++q) stage_piece(q, kb, smem);
    if constexpr (NSTAGE > 2) {   // ... and the next NSTAGE - 2 K-tiles behind it
#pragma unroll
      for (int d = 1; d < NSTAGE - 1; ++d)
        if (kb + d < ke) {
#pragma unroll
          for (int q = 0; q < PIECES; ++q) stage_piece(q, kb + d, smem + d * STAGE);
        }
    }

    // Folded LayerNorm (consumer side): (rstd, mean * rstd) of the TM rows this lane will finish (ln_finalize's table).
    // Loaded here, behind the first K-tile's DMA, so the round trip hides under the main loop (fetched in the epilogue it
    // cost ~3.5 us per band).
    float ln_rs[TM], ln_mrs[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) { ln_rs[i] = 1.f; ln_mrs[i] = 0.f; }
    if constexpr (!F8 && (EPI == GEMM_EPI_BIAS_BF16 || EPI == GEMM_EPI_GELU_BF16 || EPI == GEMM_EPI_SWIGLU_BF16)) {
      if (a.ln_stats) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const float2 st = a.ln_stats[m0 + wm * (BM / WM) + i * 32 + l31];
          ln_rs[i] = st.x;
          ln_mrs[i] = st.y;
          if constexpr (H16) {  // W (and with it acc and colsum) carries the matrix's power-of-two scale: rstd (acc - mean colsum) / s_w, exactly
            ln_rs[i] *= a.acc_scale;
            ln_mrs[i] *= a.acc_scale;
          }
        }
      }
    }

    {
      // ---- main loop: one barrier per K-tile, the next tile's DMA issued in four slices ahead of each k-step's MFMAs
      for (int t = kb; t < ke; ++t) {
        const int cur = NSTAGE == 2 ? ((t - kb) & 1) : ((t - kb) % NSTAGE);
        if constexpr (NSTAGE == 2) {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's DMA pieces of tile t have landed
        } else {
          // tiles t + 1 .. t + NSTAGE - 2 may still be in flight behind tile t (loads retire in order): PIECES instructions each.  In the tail fewer are
          // outstanding than the count would allow, so it waits for everything (conservative for the last NSTAGE - 2 tiles)
          static_assert((NSTAGE - 2) * PIECES == 12, "s_waitcnt immediate below");
          if (t + NSTAGE - 2 < ke) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();                                  // ... everyone's have; the stage refilled below (tile t - 1's) has no readers left
#ifdef FP_GEMM_TIMELINE
        if (a.dbg && t == 0) ts1 = __builtin_readcyclecounter();
#endif
        char* nxt = smem + (NSTAGE == 2 ? (cur ^ 1) : ((t - kb + NSTAGE - 1) % NSTAGE)) * STAGE;
        const bool more = t + (NSTAGE - 1) < ke;          // the tile staged during this iteration: t + NSTAGE - 1
        const char* As = smem + cur * STAGE;
        const char* Ws = As + A_BYTES;
        if constexpr (F8) {
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            if (more) {
#pragma unroll
              for (int q = 0; q < PIECES / 2; ++q) stage_piece(s * (PIECES / 2) + q, t + 1, nxt);
            }
            const int chunk = s * 4 + kh * 2;
            __builtin_amdgcn_iglp_opt(1);  // as in the bf16 loop below (+0.5 % on the fp8 pipeline)
            i32x8 af[TM], wf[TN];
            auto frag8 = [&](const char* base, int row) {
              const i32x4 lo = __builtin_bit_cast(i32x4, read_frag(base, row, chunk));
              const i32x4 hi = __builtin_bit_cast(i32x4, read_frag(base, row, chunk + 1));
              return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            };
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = frag8(As, wm * (BM / WM) + i * 32 + l31);
#pragma unroll
            for (int j = 0; j < TN; ++j) wf[j] = frag8(Ws, wn * (BN / WN) + j * 32 + l31);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
              for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wf[j], af[i], acc[i][j], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
          }
        } else if constexpr (SP && SX) {
          // f16f8 rows: an even 128-B K-tile holds the fp16 high halves of 64 k (four 16-wide hi*hi steps), the odd one their e4m3 copies
          // [hi8 x 64 | lo8 x 64] (two 64-wide fp8 MFMAs: lo_w * hi_a and hi_w * lo_a).  Two tiles per loop iteration, straight-line (a run-time
          // branch on the tile parity made the allocator spill 420 VGPRs): t is even here, its tile sits in stage 0, the fp8 tile t + 1 in stage 1.
          {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
              for (int q = (ks * PIECES) / 4; q < ((ks + 1) * PIECES) / 4; ++q) stage_piece(q, t + 1, nxt);
              const int chunk = ks * 2 + kh;
              __builtin_amdgcn_iglp_opt(1);
              f16x8 ah[TM], wh[TN];
#pragma unroll
              for (int i = 0; i < TM; ++i) ah[i] = __builtin_bit_cast(f16x8, read_frag(As, wm * (BM / WM) + i * 32 + l31, chunk));
#pragma unroll
              for (int j = 0; j < TN; ++j) wh[j] = __builtin_bit_cast(f16x8, read_frag(Ws, wn * (BN / WN) + j * 32 + l31, chunk));
#pragma unroll
              for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[j], ah[i], acc[i][j], 0, 0, 0);
            }
          }
          {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            const char* As1 = smem + STAGE;
            const char* Ws1 = As1 + A_BYTES;
            const bool more1 = t + 2 < ke;
#pragma unroll
            for (int s = 0; s < 2; ++s) {  // s = 0: lo_w * hi_a, s = 1: hi_w * lo_a
              if (more1) {
#pragma unroll
                for (int q = 0; q < PIECES / 2; ++q) stage_piece(s * (PIECES / 2) + q, t + 2, smem);
              }
              const int ca = (s == 0 ? 0 : 4) + kh * 2, cw = (s == 0 ? 4 : 0) + kh * 2;
              __builtin_amdgcn_iglp_opt(1);
              i32x8 af[TM], wf[TN];
              auto frag8 = [&](const char* base, int row, int chunk) {
                const i32x4 lo = __builtin_bit_cast(i32x4, read_frag(base, row, chunk));
                const i32x4 hi = __builtin_bit_cast(i32x4, read_frag(base, row, chunk + 1));
                return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
              };
#pragma unroll
              for (int i = 0; i < TM; ++i) af[i] = frag8(As1, wm * (BM / WM) + i * 32 + l31, ca);
#pragma unroll
              for (int j = 0; j < TN; ++j) wf[j] = frag8(Ws1, wn * (BN / WN) + j * 32 + l31, cw);
#pragma unroll
              for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                  acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wf[j], af[i], acc[i][j], 0, 0, 0, (int)FP_SX_MFMA_SCALE, 0, 0x7f7f7f7f);
            }
            ++t;
          }
        } else if constexpr (SP) {
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2) {  // two 16-wide k-steps of the 32 k-values of this tile
            if (more) {
#pragma unroll
              for (int q = 0; q < PIECES / 2; ++q) stage_piece(s2 * (PIECES / 2) + q, t + 1, nxt);
            }
            const int ch = s2 * 2 + kh;  // hi chunk of this lane's 8 k-values; the lo chunk sits 4 chunks (64 B) further
            __builtin_amdgcn_iglp_opt(1);  // as in the bf16 loop below: +1 % on the f16x3 pipeline (strategy 0: -0.7 %)
            f16x8 ah[TM], al[TM], wh[TN], wl[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
              const int row = wm * (BM / WM) + i * 32 + l31;
              ah[i] = __builtin_bit_cast(f16x8, read_frag(As, row, ch));
              al[i] = __builtin_bit_cast(f16x8, read_frag(As, row, ch + 4));
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
              const int row = wn * (BN / WN) + j * 32 + l31;
              wh[j] = __builtin_bit_cast(f16x8, read_frag(Ws, row, ch));
              wl[j] = __builtin_bit_cast(f16x8, read_frag(Ws, row, ch + 4));
            }
            // the two cross terms first (small), then hi*hi; TM*TN independent accumulators between two MFMAs of one chain
            // (FP_SP_ABLATE = 1 / 2: measurement builds that DROP one / both cross terms -- wrong results, same operand bytes -- to bound what
            //  cross terms on the half-cost fp8 pipe could buy: tools/build_variant.sh, profiles/EXPERIMENTS.md round 5; never in the shipped library)
#if !defined(FP_SP_ABLATE)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
              for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[j], ah[i], acc[i][j], 0, 0, 0);
#endif
#if !defined(FP_SP_ABLATE) || FP_SP_ABLATE < 2
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
              for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[j], al[i], acc[i][j], 0, 0, 0);
#endif
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
              for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[j], ah[i], acc[i][j], 0, 0, 0);
          }
        } else {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          if constexpr (ALT) {  // wave row 0 issues its pieces during k-steps 0, 1 -- row 1 during 2, 3
            const int slot = ks - 2 * wm;
            if (more && slot >= 0 && slot < 2) {
#pragma unroll
              for (int q = 0; q < PIECES / 2; ++q) stage_piece(slot * (PIECES / 2) + q, t + 1, nxt);
            }
          } else
          if (more) {
#pragma unroll
            for (int q = (ks * PIECES) / 4; q < ((ks + 1) * PIECES) / 4; ++q) stage_piece(q, t + (NSTAGE - 1), nxt);
          }
          const int chunk = ks * 2 + kh;
          // the compiler's MFMA / LDS-read interleaving strategy 1 for this scheduling region: +0.75 % on the pipeline in same-box A/B
          // runs on two boxes (strategy 0: -0.4 %, 2 and 3: -1.3 %; its other list-scheduling strategies: 0...-1 %)
          if constexpr (TM * TN >= 4) __builtin_amdgcn_iglp_opt(1);   // (the 64 x 128 tile's two MFMAs per k-step: the strategy's search does not terminate in reasonable memory)
          bf16x8 af[TM], wf[TN];
#pragma unroll
          for (int i = 0; i < TM; ++i) af[i] = read_frag(As, wm * (BM / WM) + i * 32 + l31, chunk);
#pragma unroll
          for (int j = 0; j < TN; ++j) wf[j] = read_frag(Ws, wn * (BN / WN) + j * 32 + l31, chunk);
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[i][j] = mfma_h<H16>(wf[j], af[i], acc[i][j]);
        }
      }
        }
      __syncthreads();
    }
#ifdef FP_GEMM_TIMELINE
    if (a.dbg) ts2 = __builtin_readcyclecounter();
#endif

    // ---- epilogue: acc[tm][tn][r] = C[m][n],  m = m0 + wm*(BM/WM) + tm*32 + (lane&31),
    //      n = n0 + wn*(BN/WN) + tn*32 + 8*(r>>2) + 4*(lane>>5) + (r&3)
    // A lane owns pieces of 32 different rows, so storing straight from registers makes every store instruction touch
    // 32 cache lines (the address path, not HBM, then bounds the tail).  Instead each 32-row band of the tile goes
    // through LDS (free after the main loop) and leaves as whole rows: 16 B per lane, lane-contiguous.
    // bf16 outputs leave through an LDS slab (whole-row 16-B stores: the tail drops from ~15k to ~8k cycles per tile);
    // so does the fp32 LayerScale + residual read-modify-write (residual rows read and written as whole rows:
    // proj 152 -> 141 us, fc2 386 -> 374 us).  Only the small fp32 bias / patch-embed outputs stay register-direct.
    constexpr bool USE_SLAB = EPI == GEMM_EPI_BIAS_BF16 || EPI == GEMM_EPI_GELU_BF16 ||
                              EPI == GEMM_EPI_SWIGLU_BF16 || RESID;
    static_assert(!F8 || USE_SLAB, "the fp8 kernels exist for the slab epilogues only");
    if constexpr (USE_SLAB) {
    constexpr bool OUT_F32 = RESID || EPI == GEMM_EPI_TOKENS_F32 || EPI == GEMM_EPI_BIAS_F32;
    constexpr int ESZ = OUT_F32 ? 4 : (F8OUT ? 1 : (SPOUT ? 4 : 2));  // split rows: hi + lo half per column
    constexpr int OUT_COLS = EPI == GEMM_EPI_SWIGLU_BF16 ? BN / 2 : BN;  // SwiGLU folds column pairs
    constexpr int SLAB_ROWS = WM * 32, SLAB_STRIDE = OUT_COLS * ESZ + 16;  // +16 B: de-phases the rows across LDS banks
    constexpr int CHUNKS_PER_ROW = OUT_COLS * ESZ / 16, SLAB_CHUNKS = SLAB_ROWS * CHUNKS_PER_ROW, NT = NW * 64;
    static_assert(SLAB_ROWS * SLAB_STRIDE <= 2 * STAGE, "slab must fit the main-loop LDS");
    // two slabs, used alternately, when they fit: band tm+1 is written while band tm is still being stored, and the
    // barrier that publishes band tm+1 also retires the readers of band tm-1's slab -> one barrier per band, not two
    constexpr bool TWO_SLABS = 2 * SLAB_ROWS * SLAB_STRIDE <= 2 * STAGE;
    constexpr int SLAB_BYTES = SLAB_ROWS * SLAB_STRIDE;
    // LayerNorm folded into this GEMM (bf16 only): A is the raw residual stream in bf16, W carries the gain, and the
    // epilogue applies out = rstd_r * (acc - mean_r * colsum_n) + bias_n before the non-linearity
    constexpr bool LN_FOLD_OK = !F8 && (EPI == GEMM_EPI_BIAS_BF16 || EPI == GEMM_EPI_GELU_BF16 || EPI == GEMM_EPI_SWIGLU_BF16);
    const bool fold = LN_FOLD_OK && a.ln_stats != nullptr;
    float4 bias[TN][4], gam[TN][4];
  #pragma unroll
    for (int tn = 0; tn < TN; ++tn)
  #pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = n0 + wn * (BN / WN) + tn * 32 + 8 * g + 4 * kh;
        bias[tn][g] = *reinterpret_cast<const float4*>(a.bias + n);
        if constexpr (EPI == GEMM_EPI_LS_RESID_F32 || F8) gam[tn][g] = *reinterpret_cast<const float4*>(a.gamma + n);
        if constexpr (LN_FOLD_OK) {
          if (fold) gam[tn][g] = *reinterpret_cast<const float4*>(a.colsum + n);  // (gam is free in these epilogues)
        }
      }
    __syncthreads();  // every wave is done with the operand tiles in LDS
    float sat_amax = 0.f;  // largest |scale * value| of a LIVE row packed into a split-fp16 / e4m3 output (saturation report)
  #pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      const int m = m0 + wm * (BM / WM) + tm * 32 + l31;
      float band_amax = 0.f;
      // Global reads of this band (residual / pos-embed rows, whole rows, all passes) are issued FIRST, ahead of the
      // register -> slab pass and its barrier, so their latency hides under that pass (proj -4 %, fc2 -1.5 %).
      // Issuing them one band ahead, interleaved with the previous band's stores, measured 5 % SLOWER.
      constexpr int PASSES = (SLAB_CHUNKS + NT - 1) / NT;
      static_assert(SLAB_CHUNKS % NT == 0, "slab chunks must divide evenly over the block");
      float4 ext[PASSES];
      int orow_i[PASSES];  // output row (< 2^31), -1: a padding row (kept as one 32-bit value per pass: the epilogue is register-bound)
      // (hi, lo) stream: a thread owns EIGHT consecutive columns per pass -- one 16-B load and one 16-B store per array, four memory
      // instructions per 8 elements where the fp32 form issues six (the tail of a residual tile is bound by its memory instructions,
      // not by its bytes: with 4-column chunks and 8-B accesses the pair measured 0.8 % slower than fp32 + bf16 copy, 1072 vs 1081)
      constexpr int CPR8 = HILO ? OUT_COLS / 8 : 1, PASS8 = HILO ? SLAB_ROWS * CPR8 / NT : 1;
      uint4 exh[PASS8], exl[PASS8];
      if constexpr (HILO) {
        static_assert(!HILO || (SLAB_ROWS * CPR8) % NT == 0, "slab chunks must divide evenly over the block");
  #pragma unroll
        for (int it = 0; it < PASS8; ++it) {
          const int id = tid + it * NT;
          const int r = id / CPR8, c = id - r * CPR8;
          const int gm_row = m0 + (r >> 5) * (BM / WM) + tm * 32 + (r & 31);
          const bool okr = gm_row < a.M_valid;
          if (okr) {
            exh[it] = *reinterpret_cast<const uint4*>(a.xb + (size_t)gm_row * a.ld_xb + n0 + c * 8);
            exl[it] = *reinterpret_cast<const uint4*>(a.xl + (size_t)gm_row * a.ld_xb + n0 + c * 8);
          }
          orow_i[it] = okr ? gm_row : -1;
        }
      }
  #pragma unroll
      for (int it = 0; it < (HILO ? 0 : PASSES); ++it) {
        const int id = tid + it * NT;
        const int r = id / CHUNKS_PER_ROW, c = id - r * CHUNKS_PER_ROW;
        const int gm_row = m0 + (r >> 5) * (BM / WM) + tm * 32 + (r & 31);
        const bool okr = gm_row < a.M_valid;
        int orow = gm_row;
        if constexpr (EPI == GEMM_EPI_TOKENS_F32) {
          const int b = gm_row / a.tok_np, pidx = gm_row - b * a.tok_np;
          orow = b * a.tok_n + a.tok_skip + pidx;
          if (okr) ext[it] = *reinterpret_cast<const float4*>(a.pos + (size_t)pidx * a.ldo + n0 + c * 4);
        }
        if constexpr (RESID) {
          if (okr) ext[it] = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(a.out) + (size_t)orow * a.ldo + n0 + c * 4);
        }
        orow_i[it] = okr ? orow : -1;
      }
      const float rs = ln_rs[tm], mrs = ln_mrs[tm];  // folded LayerNorm: rstd of this lane's row and mean * rstd
      // (a) registers -> slab (final values except for the operand that needs a global read)
      char* slab = smem + (TWO_SLABS ? (tm & 1) * SLAB_BYTES : 0);
      char* srow = slab + (wm * 32 + l31) * SLAB_STRIDE;
  #pragma unroll
      for (int tn = 0; tn < TN; ++tn)
  #pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int col = wn * (BN / WN) + tn * 32 + 8 * g + 4 * kh;
          const float4 bs = bias[tn][g];
          float v0 = acc[tm][tn][4 * g + 0] + bs.x, v1 = acc[tm][tn][4 * g + 1] + bs.y;
          float v2 = acc[tm][tn][4 * g + 2] + bs.z, v3 = acc[tm][tn][4 * g + 3] + bs.w;
          if constexpr (SP || H16) {  // undo the power-of-two operand scales (exact), then the bias  (H16: the weight matrix's scale, 1 when the caller gave none)
            const float as = a.acc_scale;
            v0 = fmaf(acc[tm][tn][4 * g + 0], as, bs.x); v1 = fmaf(acc[tm][tn][4 * g + 1], as, bs.y);
            v2 = fmaf(acc[tm][tn][4 * g + 2], as, bs.z); v3 = fmaf(acc[tm][tn][4 * g + 3], as, bs.w);
          }
          if constexpr (LN_FOLD_OK) {
            if (fold) {  // rstd * acc - (mean * rstd) * colsum + bias, two columns per v_pk_fma_f32
              const float4 cs = gam[tn][g];
              const f32x2 rs2 = {rs, rs}, nm2 = {-mrs, -mrs};
              const f32x2 t01 = __builtin_elementwise_fma(nm2, f32x2{cs.x, cs.y}, f32x2{bs.x, bs.y});
              const f32x2 t23 = __builtin_elementwise_fma(nm2, f32x2{cs.z, cs.w}, f32x2{bs.z, bs.w});
              const f32x2 r01 = __builtin_elementwise_fma(rs2, f32x2{acc[tm][tn][4 * g + 0], acc[tm][tn][4 * g + 1]}, t01);
              const f32x2 r23 = __builtin_elementwise_fma(rs2, f32x2{acc[tm][tn][4 * g + 2], acc[tm][tn][4 * g + 3]}, t23);
              v0 = r01[0]; v1 = r01[1]; v2 = r23[0]; v3 = r23[1];
            }
          }
          if constexpr (F8 && EPI != GEMM_EPI_LS_RESID_F32) {  // dequantise before the non-linearity
            const float4 gm = gam[tn][g];
            v0 *= gm.x; v1 *= gm.y; v2 *= gm.z; v3 *= gm.w;
          }
          if constexpr (EPI == GEMM_EPI_GELU_BF16) {
#ifndef FP_F16_GELU_ERF   // (-DFP_F16_GELU_ERF, measurement build: the erf form in the fp16 epilogue -- fc1 468 instead of 426 us, the same index agreement)
            if constexpr (H16) {
              const f32x2 g01 = gelu_pk9(f32x2{v0, v1}), g23 = gelu_pk9(f32x2{v2, v3});
              v0 = g01[0]; v1 = g01[1]; v2 = g23[0]; v3 = g23[1];
            } else
#endif
            if constexpr (SP || H16) {  // the erf form at fp32 accuracy: the split modes do not approximate below the arithmetic they emulate
#ifdef FP_SPLIT_GELU_OCML   // (measurement build: ocml's erff, the round-3 epilogue)
              v0 = 0.5f * v0 * (1.f + erff(v0 * 0.70710678118654752440f)); v1 = 0.5f * v1 * (1.f + erff(v1 * 0.70710678118654752440f));
              v2 = 0.5f * v2 * (1.f + erff(v2 * 0.70710678118654752440f)); v3 = 0.5f * v3 * (1.f + erff(v3 * 0.70710678118654752440f));
#else
              const f32x2 g01 = gelu_erf_pk(f32x2{v0, v1}), g23 = gelu_erf_pk(f32x2{v2, v3});
              v0 = g01[0]; v1 = g01[1]; v2 = g23[0]; v3 = g23[1];
#endif
            } else {
              const f32x2 g01 = gelu_pk(f32x2{v0, v1}), g23 = gelu_pk(f32x2{v2, v3});
              v0 = g01[0]; v1 = g01[1]; v2 = g23[0]; v3 = g23[1];
            }
          }
          if constexpr (EPI == GEMM_EPI_LS_RESID_F32) {
            const float4 gm = gam[tn][g];
            v0 *= gm.x; v1 *= gm.y; v2 *= gm.z; v3 *= gm.w;
          }
          if constexpr (EPI == GEMM_EPI_SWIGLU_BF16) {
            float h0, h1;
            if constexpr (SP) {  // exact expf form, like the fp32 path
              h0 = v0 / (1.f + expf(-v0)) * v1;
              h1 = v2 / (1.f + expf(-v2)) * v3;
            } else {
              h0 = v0 / (1.f + __builtin_amdgcn_exp2f(-v0 * 1.44269504088896340736f)) * v1;  // silu(x1) * x2
              h1 = v2 / (1.f + __builtin_amdgcn_exp2f(-v2 * 1.44269504088896340736f)) * v3;
            }
            if constexpr (SPOUT && SX) {
              unsigned hi, p8;
              splitx_pack2(h0, h1, a.out_scale, hi, p8, band_amax);
              splitx_store2(srow, col >> 1, hi, p8);
            } else if constexpr (SPOUT) {
              unsigned hi, lo;
              split16_pack2(h0, h1, a.out_scale, hi, lo, band_amax);
              char* sp = srow + split16_pos(col >> 1) * 2;
              *reinterpret_cast<unsigned*>(sp) = hi;
              *reinterpret_cast<unsigned*>(sp + 64) = lo;
            } else if constexpr (F8OUT) *reinterpret_cast<unsigned short*>(srow + (col >> 1)) = (unsigned short)pack_fp8x4(h0 * a.out_scale, h1 * a.out_scale, 0.f, 0.f, band_amax);
            else *reinterpret_cast<unsigned*>(srow + (col >> 1) * 2) = pack_h2<H16>(h0, h1);
          } else if constexpr (SPOUT && SX && EPI != GEMM_EPI_BIAS_BF16) {
            unsigned h01, p01, h23, p23;
            splitx_pack2(v0, v1, a.out_scale, h01, p01, band_amax);
            splitx_pack2(v2, v3, a.out_scale, h23, p23, band_amax);
            splitx_store4(srow, col, h01, p01, h23, p23);
          } else if constexpr (SPOUT) {
            unsigned h01, l01, h23, l23;
            split16_pack2(v0, v1, a.out_scale, h01, l01, band_amax);
            split16_pack2(v2, v3, a.out_scale, h23, l23, band_amax);
            char* sp = srow + split16_pos(col) * 2;
            *reinterpret_cast<uint2*>(sp) = make_uint2(h01, h23);
            *reinterpret_cast<uint2*>(sp + 64) = make_uint2(l01, l23);
          } else if constexpr (OUT_F32) *reinterpret_cast<float4*>(srow + col * 4) = make_float4(v0, v1, v2, v3);
          else if constexpr (F8OUT) *reinterpret_cast<unsigned*>(srow + col) = pack_fp8x4(v0 * a.out_scale, v1 * a.out_scale, v2 * a.out_scale, v3 * a.out_scale, band_amax);
          else *reinterpret_cast<uint2*>(srow + col * 2) = make_uint2(pack_h2<H16>(v0, v1), pack_h2<H16>(v2, v3));
        }
      if constexpr (SPOUT || F8OUT) {
        if (m < a.M_valid) sat_amax = nanmax3(sat_amax, band_amax, 0.f);  // padding rows (computed, never stored) do not report
      }
      __syncthreads();
      // (b) slab -> global, whole rows
      if constexpr (HILO) {
  #pragma unroll
        for (int it = 0; it < PASS8; ++it) {
          const int id = tid + it * NT;
          const int r = id / CPR8, c = id - r * CPR8;
          if (orow_i[it] < 0) continue;
          const size_t orow_it = (size_t)orow_i[it];
          const char* sp = slab + r * SLAB_STRIDE + c * 32;
          const float4 s0 = *reinterpret_cast<const float4*>(sp), s1v = *reinterpret_cast<const float4*>(sp + 16);
          const unsigned hw[4] = {exh[it].x, exh[it].y, exh[it].z, exh[it].w}, lw[4] = {exl[it].x, exl[it].y, exl[it].z, exl[it].w};
          float v[8] = {s0.x, s0.y, s0.z, s0.w, s1v.x, s1v.y, s1v.z, s1v.w};
          unsigned ho[4], lo[4];
          float s1 = 0.f, s2 = 0.f;
  #pragma unroll
          for (int q = 0; q < 4; ++q) {   // x = hi + lo (exact in fp32: lo lies within 2^-9 of hi's last place), x' = x + (acc + bias)
            const f32x2 xh = unpack_h2<H16>(hw[q]), xlo = unpack_h2<H16>(lw[q]);
            v[2 * q] += xh[0] + xlo[0];
            v[2 * q + 1] += xh[1] + xlo[1];
            ho[q] = pack_h2<H16>(v[2 * q], v[2 * q + 1]);
            const f32x2 nh = unpack_h2<H16>(ho[q]);
            lo[q] = pack_h2<H16>(v[2 * q] - nh[0], v[2 * q + 1] - nh[1]);
            s1 += v[2 * q] + v[2 * q + 1];
            s2 += fmaf(v[2 * q], v[2 * q], v[2 * q + 1] * v[2 * q + 1]);
          }
          *reinterpret_cast<uint4*>(a.xb + orow_it * a.ld_xb + n0 + c * 8) = make_uint4(ho[0], ho[1], ho[2], ho[3]);
          *reinterpret_cast<uint4*>(a.xl + orow_it * a.ld_xb + n0 + c * 8) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
          // partial sums over 128-column groups = 16 lanes x 8 columns; the same tree whatever the tile width
          s1 = row16_sum(s1);
          s2 = row16_sum(s2);
          if ((c & 15) == 15) a.stats_out[(size_t)(n0 / 128 + (c >> 4)) * a.M + orow_it] = make_float2(s1, s2);
        }
      }
  #pragma unroll
      for (int it = 0; it < (HILO ? 0 : PASSES); ++it) {
        const int id = tid + it * NT;
        const int r = id / CHUNKS_PER_ROW, c = id - r * CHUNKS_PER_ROW;
        if (orow_i[it] < 0) continue;
        const size_t orow_it = (size_t)orow_i[it];
        const char* sp = slab + r * SLAB_STRIDE + c * 16;
        if constexpr (F8OUT) {
          const int ncol = (EPI == GEMM_EPI_SWIGLU_BF16 ? n0 / 2 : n0) + c * 16;
          *reinterpret_cast<uint4*>(reinterpret_cast<unsigned char*>(a.out) + orow_it * a.ldo + ncol) = *reinterpret_cast<const uint4*>(sp);
        } else if constexpr (!OUT_F32) {
          const int ncol = (EPI == GEMM_EPI_SWIGLU_BF16 ? n0 / 2 : n0) * (SPOUT ? 2 : 1) + c * 8;  // split rows: 2 halves per column
          *reinterpret_cast<uint4*>(reinterpret_cast<__bf16*>(a.out) + orow_it * a.ldo + ncol) = *reinterpret_cast<const uint4*>(sp);
        } else {
          float4 v = *reinterpret_cast<const float4*>(sp);
          if constexpr (RESID || EPI == GEMM_EPI_TOKENS_F32) {
            v.x += ext[it].x; v.y += ext[it].y; v.z += ext[it].z; v.w += ext[it].w;
          }
          *reinterpret_cast<float4*>(reinterpret_cast<float*>(a.out) + orow_it * a.ldo + n0 + c * 4) = v;
          if constexpr (EPI == GEMM_EPI_RESID_F32 && !F8) {
            if (a.xb) {  // the next GEMM's A operand + this tile's share of the row's LayerNorm statistics
              *reinterpret_cast<uint2*>(a.xb + orow_it * a.ld_xb + n0 + c * 4) = make_uint2(pack_h2<H16>(v.x, v.y), pack_h2<H16>(v.z, v.w));
              // partial sums over 128-column groups -- 32 lanes x float4, the same tree whatever the tile width, so a row's
              // statistics (and everything downstream) do not depend on which tile shape the batch size selects.  DPP adds
              // (VALU rate): the ds_bpermute chain of __shfl_xor cost 24 k cycles per tile here.
              float s1 = (v.x + v.y) + (v.z + v.w), s2 = fmaf(v.x, v.x, v.y * v.y) + fmaf(v.z, v.z, v.w * v.w);
              s1 = row32_sum(s1);
              s2 = row32_sum(s2);
              if ((c & 31) == 31) a.stats_out[(size_t)(n0 / 128 + (c >> 5)) * a.M + orow_it] = make_float2(s1, s2);
            }
          }
        }
      }
      if (!TWO_SLABS && tm + 1 < TM) __syncthreads();
    }
    if constexpr (SPOUT) report_saturation(a.sat, 0, sat_amax, (SX && EPI != GEMM_EPI_BIAS_BF16) ? FP_SX_MAX : FP_F16_MAX);   // (f16f8 output rows: the e4m3 copy's range, common.hpp)
    if constexpr (F8OUT) report_saturation(a.sat, 1, sat_amax, FP_E4M3_MAX);
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
      int pidx = 0;
      if constexpr (EPI == GEMM_EPI_TOKENS_F32) {
        const int b = m / a.tok_np;
        pidx = m - b * a.tok_np;
        out_row = (size_t)b * a.tok_n + a.tok_skip + pidx;
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
          if constexpr (SP || H16) {
            const float as = a.acc_scale;
            v0 = fmaf(acc[tm][tn][4 * g + 0], as, bs.x); v1 = fmaf(acc[tm][tn][4 * g + 1], as, bs.y);
            v2 = fmaf(acc[tm][tn][4 * g + 2], as, bs.z); v3 = fmaf(acc[tm][tn][4 * g + 3], as, bs.w);
          }
          if constexpr (EPI == GEMM_EPI_BIAS_BF16 || EPI == GEMM_EPI_GELU_BF16) {
            if constexpr (EPI == GEMM_EPI_GELU_BF16) {
              const f32x2 g01 = gelu_pk(f32x2{v0, v1}), g23 = gelu_pk(f32x2{v2, v3});
              v0 = g01[0]; v1 = g01[1]; v2 = g23[0]; v3 = g23[1];
            }
            uint2 pk = make_uint2(pack_h2<H16>(v0, v1), pack_h2<H16>(v2, v3));
            *reinterpret_cast<uint2*>(reinterpret_cast<__bf16*>(a.out) + out_row * a.ldo + n) = pk;
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
#ifdef FP_GEMM_TIMELINE
    if (a.dbg) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const unsigned long long ts3 = __builtin_readcyclecounter();
      if (tid == 0) {
        unsigned long long* d = a.dbg + (size_t)blockIdx.x * 4;
        d[0] = ts0; d[1] = ts1; d[2] = ts2; d[3] = ts3;
      }
    }
#endif
  }
}

// The super-tile raster of a launch (0 x 0: none -- row-major tile ids in one contiguous chunk per XCD).
// FP_GEMM_RAST (FP_EXPERIMENTS builds only): 0 = off, 1 = the default below, "RxG" = another super-tile of R m-tiles x G n-tiles (R * G = 32; measurements:
// profiles/EXPERIMENTS.md "super-tile shapes" -- 4x8 / 2x16 / 16x2 all lose to 8x4: only a 4-n-tile W panel (2 MiB) survives in a 4-MiB L2
// next to the streaming A slab, and the A re-fetch per n-group that remains is what a wider group would remove).
// Default: 4 x 8 when the output is a multiple of 8 n-tiles wide (fc1: 16), else 8 x 4 (qkv: 12) -- same-box pipeline A/B, three
// alternations: 1043.0 detections/s against 1038.2 with 8 x 4 everywhere (profiles/EXPERIMENTS.md)
struct GemmRaster { int r, gn; };
static GemmRaster pick_raster(int bm, int n_tiles, unsigned grid) {
#ifdef FP_EXPERIMENTS   // (measurement builds: the super-tile shape of the sweep, read once)
  static const int env_rast = getenv("FP_GEMM_RAST") ? atoi(getenv("FP_GEMM_RAST")) : 1;
  static const int env_gn = (getenv("FP_GEMM_RAST") && strchr(getenv("FP_GEMM_RAST"), 'x')) ? atoi(strchr(getenv("FP_GEMM_RAST"), 'x') + 1) : 0;
#else
  constexpr int env_rast = 1, env_gn = 0;
#endif
  const int wide8 = n_tiles % 8 == 0;
  const int rr = env_gn ? env_rast : (wide8 ? 4 : 8), gn = env_gn ? env_gn : (wide8 ? 8 : 4);
  if (env_rast && rr * gn == 32 && bm >= 256 && n_tiles % gn == 0 && n_tiles > 4 && grid >= 512) return {rr, gn};
  return {0, 0};
}

template <int EPI, int BM, int BN, int WM, int WN, bool F8 = false, bool F8OUT = false, bool SP = false, bool SPOUT = false, bool SX = false, bool H16 = false, int NSTAGE = 2>
int launch_cfg(const GemmBf16Args& a_in, hipStream_t st) {
  GemmBf16Args a = a_in;
  if constexpr (H16) {
    if (!(a.acc_scale > 0.f)) a.acc_scale = 1.f;   // fp16 weights without a scale (the plain entry points)
  }
  // M is padded to whole tiles of every shape in use; tiles of padding rows only are not launched (they would all sit at the end of the
  // tile order, i.e. in the last XCD's chunk, and leave that XCD short of work)
  a.m_tiles = a.M / BM;
  if (a.M_valid > 0 && (a.M_valid + BM - 1) / BM < a.m_tiles) a.m_tiles = (a.M_valid + BM - 1) / BM;
  unsigned grid = a.m_tiles * (a.N / BN);
  a.rast_r = a.rast_gn = 0;
  // Tile order: by default row-major tile ids cut into one contiguous chunk per XCD.  For wide outputs (more than 4
  // n-tiles: fc1, qkv) a super-tile raster instead: an XCD's 32 concurrent workgroups form 8 m-tiles x 4 n-tiles and
  // walk down M inside one group of 4 n-tiles, so each W K-slice is shared by 8 workgroups (A by 4) and the group's W
  // panel stays in the XCD's L2: L2 hit rate 64 -> 75 %, fc1 412 -> 399 us.  (N = 1024 is 4 n-tiles wide: the default
  // order already has that shape, and the raster's intra-order measured 4 % slower there.)
  const GemmRaster ra = pick_raster(BM, a.N / BN, grid);
  if (ra.r) {
    a.rast_r = ra.r; a.rast_gn = ra.gn;
    grid = ((a.m_tiles + ra.r - 1) / ra.r) * ((a.N / BN) / ra.gn) * 32;
  }
  const size_t lds = (size_t)(BM + BN) * BK * 2 * NSTAGE;
  // (Measured and dropped: asking for > 80 KiB of LDS when a launch has no more tiles than CUs, so that every workgroup takes a CU of its own -- the dispatcher
  //  already spreads them: fc2 of a two-crop batch 51.4 us either way, tools/b1_gemm_probe.py.)
  static FpDeviceOnce attr;
  fp_allow_dynamic_lds(attr, &gemm_bf16_kernel<EPI, BM, BN, WM, WN, F8, F8OUT, SP, SPOUT, SX, H16, NSTAGE>, (int)lds);
  hipLaunchKernelGGL((gemm_bf16_kernel<EPI, BM, BN, WM, WN, F8, F8OUT, SP, SPOUT, SX, H16, NSTAGE>), dim3(grid), dim3(WM * WN * 64), lds, st, a);
  FP_CHECK_LAUNCH("gemm_bf16_kernel");
  return FP_OK;
}

// 320 x 256 tiles: 160 accumulator registers per lane (<= 256 VGPRs, no spill), 10 % fewer operand bytes through the L1 fill path per flop --
// the path that bounds the main loop -- and another round count.  Same k order per output element: results bit-identical to the 256^2 tile's.
// Chosen when M is a whole number of both tile heights (the extractor pads to 1280 rows when that is cheap) and the round count favours it.
// A launch runs in XCD rounds: an XCD's 32 CUs take one super-tile of the raster (32 tiles; ragged super-rows leave holes) or the next 32
// tiles of the XCD's chunk per round, and a partly filled last round costs a whole one -- a sweep over 18 batch sizes (profiles/EXPERIMENTS.md
// "320 x 256 block tiles") shows steps exactly at these counts: qkv at the bench batch, 256^2: 22 x 3 = 66 super-tiles = 9 rounds of 31 us;
// 320 x 256: 18 x 3 = 54 = 7 rounds of 37.5 us.  A round of the taller tile takes 1.21x (not 1.25x) the time: cost = rounds x height x 0.97.
// The estimate picks the faster tile in 32 of the 36 cases of the recorded sweep (profiles/r4_gemm_tile_sweep.txt; three misses within 1 %, one
// of 3 %) and 256 rows for the residual GEMMs of the bench batch (3 rounds either way).  GemmBf16Args.no_tall (fp_vit_model.flags & FP_VIT_NO_TALL_TILES) is the A/B switch.
static int xcd_rounds(int bm, int m_valid, int n_tiles) {
  const int m_tiles = (m_valid + bm - 1) / bm, xcds = 8, per_xcd = fp_num_cus() / xcds > 0 ? fp_num_cus() / xcds : 32;
  const GemmRaster ra = pick_raster(bm, n_tiles, (unsigned)(m_tiles * n_tiles));
  if (ra.r) return (((m_tiles + ra.r - 1) / ra.r) * (n_tiles / ra.gn) + xcds - 1) / xcds;
  const int chunk = (m_tiles * n_tiles + xcds - 1) / xcds;
  return (chunk + per_xcd - 1) / per_xcd;
}
static bool tall_tile_wins(const GemmBf16Args& a) {
  if (a.no_tall || a.M % 320 != 0 || a.M % 256 != 0 || a.N % 256 != 0) return false;
  return (float)(xcd_rounds(320, a.M_valid, a.N / 256) * 320) * 0.97f < (float)(xcd_rounds(256, a.M_valid, a.N / 256) * 256);
}

// Tile selection: 256x256 (8 waves, 1 block/CU, 128 KiB LDS) when the shape allows it and fills the chip,
// otherwise 128x128 (4 waves, 2 blocks/CU).
template <int EPI, bool SP = false, bool SPOUT = false, bool SX = false, bool H16 = false>
int launch(const GemmBf16Args& a, hipStream_t st) {
  const int force = a.tile_override;
  const bool big_ok = a.M % 256 == 0 && a.N % 256 == 0;
  // (between 1 and 1.5 rounds of 256^2 tiles the second round is mostly idle CUs and the 128^2 tiles win: 308 tiles of the
  //  hooked block's selected rows, proj 79 -> 67 us, fc2 207 -> 195 us; results do not depend on the tile)
  const int tiles_big = (a.M / 256) * (a.N / 256), cus = fp_num_cus();
  const bool use_big = big_ok && (force == 256 || (force == 0 && tiles_big >= cus && !(tiles_big > cus && tiles_big < cus + cus / 2)));
  // (Measured and dropped, round 3: sending the m-tiles that hold the few tiles beyond a whole number of rounds -- qkv at the bench batch:
  //  2064 = 8 x 256 + 16 -- as 128^2 tiles in a second launch: 334 vs 289 us; a dependent second launch costs its own latency.)
  if constexpr (!SP && (EPI == GEMM_EPI_BIAS_BF16 || EPI == GEMM_EPI_GELU_BF16 || EPI == GEMM_EPI_RESID_HILO)) {
    if ((force == 320 && a.M % 320 == 0 && a.N % 256 == 0) || (force == 0 && use_big && tall_tile_wins(a)))
      return launch_cfg<EPI, 320, 256, 2, 4, false, false, SP, SPOUT, false, H16>(a, st);
  }
  if (use_big) return launch_cfg<EPI, 256, 256, 2, 4, false, false, SP, SPOUT, SX, H16>(a, st);
  // Small M (a batch of one or two crops -- the reference loop's shape, one detection at a time, scripts/infer.py:368): the N = D outputs (proj,
  // fc2) are 12 x 8 = 96 tiles of 128^2 at B = 1 and leave 160 of the 256 CUs idle through fc2's 64 K-tiles (43 us per launch, the largest
  // bucket of a B = 1 forward).  64 x 128 tiles double the count; same k order per output element -> the same bits.
  if constexpr (!SP && (EPI == GEMM_EPI_RESID_HILO || EPI == GEMM_EPI_RESID_F32 || EPI == GEMM_EPI_LS_RESID_F32)) {
    const int tiles_128 = (a.M_valid > 0 ? (a.M_valid + 127) / 128 : a.M / 128) * (a.N / 128);
    if ((force == 64 && a.M % 64 == 0) || (force == 0 && a.M % 64 == 0 && tiles_128 <= cus / 2))
#ifdef FP_GEMM_SMALL_2STAGE   // (measurement build: the double-buffered loop in the small tile)
      return launch_cfg<EPI, 64, 128, 2, 2, false, false, SP, SPOUT, false, H16, 2>(a, st);
#endif
      return launch_cfg<EPI, 64, 128, 2, 2, false, false, SP, SPOUT, false, H16, 4>(a, st);   // four K-tiles in flight: one workgroup per CU has nothing else to hide the fetch latency behind
  }
  return launch_cfg<EPI, 128, 128, 2, 2, false, false, SP, SPOUT, SX, H16>(a, st);
}

}  // namespace

// The kernel template above is instantiated by four translation units so that no single compile holds every instantiation (the one-file build
// ran out of memory): gemm_bf16.hip (bf16 operands), gemm_fp8.hip (-> FP_GEMM_TU == 2), gemm_split.hip (-> 3), gemm_splitx.hip (-> 4), gemm_f16.hip (-> 5);
// the latter four are one-line files that define FP_GEMM_TU and include this one.
#ifndef FP_GEMM_TU
#define FP_GEMM_TU 1
#endif

#if FP_GEMM_TU == 2
// fp8 (e4m3) operands: A [M, K] and W [N, K] one byte per element, K a multiple of 128, M and N multiples of 256;
// a.gamma = dequantisation scale per output column (x LayerScale for LS_RESID), a.bias already divided by it.
int gemm_fp8_launch(int epi, const GemmBf16Args& a_in, hipStream_t st) {
  GemmBf16Args a = a_in;
  FP_REQUIRE(a.M > 0 && a.M % 256 == 0 && a.N > 0 && a.N % 256 == 0, "gemm_fp8: M (%d) and N (%d) must be positive multiples of 256", a.M, a.N);
  FP_REQUIRE(a.K > 0 && a.K % 128 == 0, "gemm_fp8: K (%d) must be a multiple of 128", a.K);
  FP_REQUIRE(a.bias != nullptr && a.gamma != nullptr, "gemm_fp8: bias and the per-column scale are required");
  FP_REQUIRE(a.lda % 16 == 0 && a.ldw % 16 == 0 && a.ldo % 4 == 0, "gemm_fp8: leading dims must keep 16-byte alignment");
  a.K /= 2; a.lda /= 2; a.ldw /= 2;  // an fp8 row addressed as a bf16 row of half the length (see the kernel header)
  const bool tall = a.tile_override == 320 ? a.M % 320 == 0 : (a.tile_override == 0 && tall_tile_wins(a));   // (the residual epilogue keeps 256 rows)
  if (a.out_scale > 0.f) {  // fp8 output
    FP_REQUIRE(a.ldo % 16 == 0, "gemm_fp8: an fp8 output needs ldo %% 16 == 0");
    if (epi == GEMM_EPI_GELU_BF16) return tall ? launch_cfg<GEMM_EPI_GELU_BF16, 320, 256, 2, 4, true, true>(a, st) : launch_cfg<GEMM_EPI_GELU_BF16, 256, 256, 2, 4, true, true>(a, st);
    if (epi == GEMM_EPI_SWIGLU_BF16) return tall ? launch_cfg<GEMM_EPI_SWIGLU_BF16, 320, 256, 2, 4, true, true>(a, st) : launch_cfg<GEMM_EPI_SWIGLU_BF16, 256, 256, 2, 4, true, true>(a, st);
    fp_set_error("gemm_fp8: fp8 output exists for the GELU and SwiGLU epilogues only (epilogue %d)", epi);
    return FP_ERR_UNSUPPORTED;
  }
  switch (epi) {
    case GEMM_EPI_BIAS_BF16: return tall ? launch_cfg<GEMM_EPI_BIAS_BF16, 320, 256, 2, 4, true>(a, st) : launch_cfg<GEMM_EPI_BIAS_BF16, 256, 256, 2, 4, true>(a, st);
    case GEMM_EPI_GELU_BF16: return tall ? launch_cfg<GEMM_EPI_GELU_BF16, 320, 256, 2, 4, true>(a, st) : launch_cfg<GEMM_EPI_GELU_BF16, 256, 256, 2, 4, true>(a, st);
    case GEMM_EPI_LS_RESID_F32: return launch_cfg<GEMM_EPI_LS_RESID_F32, 256, 256, 2, 4, true>(a, st);
    case GEMM_EPI_SWIGLU_BF16: return tall ? launch_cfg<GEMM_EPI_SWIGLU_BF16, 320, 256, 2, 4, true>(a, st) : launch_cfg<GEMM_EPI_SWIGLU_BF16, 256, 256, 2, 4, true>(a, st);
  }
  fp_set_error("gemm_fp8: epilogue %d is not available for fp8 operands", epi);
  return FP_ERR_UNSUPPORTED;
}

#endif  // FP_GEMM_TU == 2

#if FP_GEMM_TU == 3 || FP_GEMM_TU == 4
// split-fp16 operands (f16x3 mode, TU 3) / f16f8 rows (f16f8 mode, TU 4): a.K is the LOGICAL K; the kernel walks rows of 2K halves
#if FP_GEMM_TU == 3
int gemm_split_launch(int epi, const GemmBf16Args& a_in, hipStream_t st) {
  constexpr bool SX = false;
#else
int gemm_splitx_launch(int epi, const GemmBf16Args& a_in, hipStream_t st) {
  constexpr bool SX = true;
#endif
  GemmBf16Args a = a_in;
  FP_REQUIRE(a.M > 0 && a.M % 128 == 0 && a.N > 0 && a.N % 128 == 0, "gemm_split: M (%d) and N (%d) must be positive multiples of 128", a.M, a.N);
  FP_REQUIRE(a.K > 0 && a.K % (SX ? 64 : 32) == 0, "gemm_split: K (%d) must be a multiple of %d", a.K, SX ? 64 : 32);
  FP_REQUIRE(a.bias != nullptr, "gemm_split: bias is required (pass zeros)");
  FP_REQUIRE(a.lda % 8 == 0 && a.ldw % 8 == 0 && a.lda >= 2 * a.K && a.ldw >= 2 * a.K, "gemm_split: operand rows are 2K halves, 16-byte aligned");
  FP_REQUIRE(a.acc_scale > 0.f, "gemm_split: acc_scale must be positive");
  const bool half_out = epi == GEMM_EPI_BIAS_BF16 || epi == GEMM_EPI_GELU_BF16 || epi == GEMM_EPI_SWIGLU_BF16;
  FP_REQUIRE(!half_out || (a.out_scale > 0.f && a.ldo % 8 == 0), "gemm_split: a split-fp16 output needs out_scale > 0 and ldo %% 8 == 0");
  FP_REQUIRE(half_out || a.ldo % 4 == 0, "gemm_split: ldo must keep 16-byte alignment");
  FP_REQUIRE(epi != GEMM_EPI_LS_RESID_F32 || a.gamma, "gemm_split: gamma required");
  a.K *= 2;  // halves per row: one 64-half K-tile = 32 logical k (f16f8: a pair of tiles = 64 logical k)
  switch (epi) {
    case GEMM_EPI_BIAS_BF16: return launch<GEMM_EPI_BIAS_BF16, true, true, SX>(a, st);
    case GEMM_EPI_GELU_BF16: return launch<GEMM_EPI_GELU_BF16, true, true, SX>(a, st);
    case GEMM_EPI_SWIGLU_BF16: return launch<GEMM_EPI_SWIGLU_BF16, true, true, SX>(a, st);
    case GEMM_EPI_LS_RESID_F32: return launch<GEMM_EPI_LS_RESID_F32, true, false, SX>(a, st);
    case GEMM_EPI_TOKENS_F32: return launch<GEMM_EPI_TOKENS_F32, true, false, SX>(a, st);
    case GEMM_EPI_BIAS_F32: return launch<GEMM_EPI_BIAS_F32, true, false, SX>(a, st);
  }
  fp_set_error("gemm_split: epilogue %d is not available for split-fp16 operands", epi);
  return FP_ERR_UNSUPPORTED;
}
#endif  // FP_GEMM_TU == 3 || 4

#if FP_GEMM_TU == 1 || FP_GEMM_TU == 5
// TU 1: bf16 operands (gemm_bf16_launch); TU 5 (gemm_f16.hip): the same kernels on IEEE fp16 operands (gemm_f16_launch, the "f16" mode)
#if FP_GEMM_TU == 1
#define L(EPI) launch<EPI>(a, st)
int gemm_bf16_launch(int epi, const GemmBf16Args& a, hipStream_t st) {
#else
#define L(EPI) launch<EPI, false, false, false, true>(a, st)
int gemm_f16_launch(int epi, const GemmBf16Args& a, hipStream_t st) {
#endif
  FP_REQUIRE(a.M > 0 && a.M % 128 == 0, "gemm_bf16: M (%d) must be a positive multiple of 128 (pad the activation buffer)", a.M);
  FP_REQUIRE(a.N > 0 && a.N % 128 == 0, "gemm_bf16: N (%d) must be a multiple of 128", a.N);
  FP_REQUIRE(a.K > 0 && a.K % BK == 0, "gemm_bf16: K (%d) must be a multiple of %d", a.K, BK);
  FP_REQUIRE(a.bias != nullptr, "gemm_bf16: bias is required (pass zeros)");
  FP_REQUIRE(a.lda % 8 == 0 && a.ldw % 8 == 0 && a.ldo % 4 == 0, "gemm_bf16: leading dims must keep 16-byte alignment");
  switch (epi) {
    case GEMM_EPI_BIAS_BF16: return L(GEMM_EPI_BIAS_BF16);
    case GEMM_EPI_GELU_BF16: return L(GEMM_EPI_GELU_BF16);
    case GEMM_EPI_LS_RESID_F32: return L(GEMM_EPI_LS_RESID_F32);
    case GEMM_EPI_RESID_F32: return L(GEMM_EPI_RESID_F32);
    case GEMM_EPI_RESID_HILO:
      FP_REQUIRE(a.xb && a.xl && a.stats_out && a.N % 128 == 0 && a.ld_xb >= a.N, "gemm_bf16: the (hi, lo) residual epilogue needs xb, xl, stats and N %% 128 == 0");
      // a lane moves 8 bf16 (16 bytes) of xb and of xl per access
      FP_REQUIRE(a.ld_xb % 8 == 0 && reinterpret_cast<uintptr_t>(a.xb) % 16 == 0 && reinterpret_cast<uintptr_t>(a.xl) % 16 == 0,
                 "gemm_bf16: the (hi, lo) residual epilogue needs 16-byte aligned xb / xl and a row stride that is a multiple of 8 elements (ld_xb = %d)", a.ld_xb);
      return L(GEMM_EPI_RESID_HILO);
    case GEMM_EPI_TOKENS_F32: return L(GEMM_EPI_TOKENS_F32);
    case GEMM_EPI_BIAS_F32: return L(GEMM_EPI_BIAS_F32);
    case GEMM_EPI_SWIGLU_BF16: return L(GEMM_EPI_SWIGLU_BF16);
  }
  fp_set_error("gemm_bf16: unknown epilogue %d", epi);
  return FP_ERR_INVALID;
}
#undef L
#endif  // FP_GEMM_TU == 1 || 5
