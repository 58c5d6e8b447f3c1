// Shared device/host helpers for the gfx950 kernels (wave64, MFMA, 160 KB LDS).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

#define FP_DEVICE __device__ __forceinline__

// ---- status codes of the C ABI (include/foundpose_amd.h)
enum : int {
  FP_OK = 0,
  FP_ERR_INVALID = 1,   // bad argument (shape / alignment / dtype)
  FP_ERR_UNSUPPORTED = 2,
  FP_ERR_HIP = 3,       // a HIP runtime call or launch failed
};

void fp_set_error(const char* fmt, ...);

#define FP_REQUIRE(cond, ...)              \
  do {                                     \
    if (!(cond)) {                         \
      fp_set_error(__VA_ARGS__);           \
      return FP_ERR_INVALID;               \
    }                                      \
  } while (0)

#define FP_CHECK_LAUNCH(name)                                              \
  do {                                                                     \
    hipError_t e__ = hipGetLastError();                                    \
    if (e__ != hipSuccess) {                                               \
      fp_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
      return FP_ERR_HIP;                                                   \
    }                                                                      \
  } while (0)

// ---- bf16 <-> f32
FP_DEVICE float bf16_to_f32(__bf16 v) { return (float)v; }

// four floats -> four OCP e4m3 bytes (round to nearest even, saturating at +-448)
FP_DEVICE float clamp448(float v) { return fminf(fmaxf(v, -448.f), 448.f); }
FP_DEVICE unsigned pack_fp8x4(float v0, float v1, float v2, float v3) {
  int w = __builtin_amdgcn_cvt_pk_fp8_f32(clamp448(v0), clamp448(v1), 0, false);
  return (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(clamp448(v2), clamp448(v3), w, true);
}
// NaN-PROPAGATING maximum (v_maximum3_f32 on gfx950; fmaxf / v_max3_f32 would drop a NaN operand): the running maxima of the saturation
// report must see a NaN activation -- the clamps below (v_med3, fminf / fmaxf) turn it into a finite operand, so the report is the only
// place it can surface.
FP_DEVICE float nanmax3(float a, float b, float c) { return __builtin_elementwise_maximum(a, __builtin_elementwise_maximum(b, c)); }
// ... and the largest magnitude that went in (saturation report: a value beyond +-448 was clamped)
FP_DEVICE unsigned pack_fp8x4(float v0, float v1, float v2, float v3, float& amax) {
  amax = nanmax3(nanmax3(amax, fabsf(v0), fabsf(v1)), fabsf(v2), fabsf(v3));  // v_maximum3_f32 with |.| modifiers
  return pack_fp8x4(v0, v1, v2, v3);
}
// Sticky saturation counters (fp_vit_workspace.sat, include/foundpose_amd.h): slot 0 counts split-fp16 clamps (|s x| > 65504),
// slot 1 e4m3 clamps (|s x| > 448).  A thread keeps the running maximum of what it packed (one VALU op per pair) and reports
// once at the end of the kernel; an atomic is issued only when something actually clamped.
constexpr float FP_F16_MAX = 65504.f, FP_E4M3_MAX = 448.f;
FP_DEVICE void report_saturation(int* sat, int slot, float amax, float limit) {
  if (sat != nullptr && !(amax <= limit)) atomicAdd(sat + slot, 1);  // amax comes from nanmax3: a NaN among the packed values reports too
}

FP_DEVICE unsigned pack_bf16x2(float lo, float hi) {
  f32x2 p = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(p, bf16x2));  // v_cvt_pk_bf16_f32 (RNE)
}

// ---- the two 16-bit operand formats of the single-pass ViT pipeline: bf16 (H16 = false: precision "bf16") and IEEE fp16 (H16 = true: precision
// "f16" -- the same MFMA rate and the same bytes, 11 significant bits instead of 8, range +-65504 instead of +-3e38).  The fp16 conversion is
// v_cvt_pk_f16_f32 (round to nearest even, subnormals kept, a value beyond the range becomes inf).  Nothing is clamped and nothing is tracked where the
// rows are produced (a running maximum in the GEMM epilogues cost 3-5 % of their launches): an inf poisons its row's residual stream and, through the
// keys and values of the next attention, every token of the image, so the LAST kernel of the pipeline (final norm / sampling) counts non-finite
// features into the saturation counter (slot 0) -- a reported batch is not usable, an unreported one never saw an overflow that mattered.
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
FP_DEVICE unsigned pack_f16x2(float lo, float hi) {
  f32x2 p = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(p, f16x2));
}
template <bool H16> FP_DEVICE unsigned pack_h2(float lo, float hi) {
  if constexpr (H16) return pack_f16x2(lo, hi);
  else return pack_bf16x2(lo, hi);
}
template <bool H16> FP_DEVICE f32x2 unpack_h2(unsigned w) {   // the two 16-bit elements of a dword as fp32 (exact)
  if constexpr (H16) return __builtin_convertvector(__builtin_bit_cast(f16x2, w), f32x2);
  else return f32x2{__uint_as_float(w << 16), __uint_as_float(w & 0xffff0000u)};
}
template <bool H16> FP_DEVICE f32x16 mfma_h(bf16x8 a, bf16x8 b, f32x16 c) {   // 32x32x16, operands as raw 16-byte fragments of either format
  if constexpr (H16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// ---- split-fp16 operands of the f16x3 mode (near-exact fp32 products on the fp16 MFMA).
// A logical fp32 row x[0..K) (K % 32 == 0) is stored as 2K halves: group g = k / 32 holds hi(x[32g .. 32g+31]) in halves
// [64g, 64g + 32) and lo(...) in [64g + 32, 64g + 64), with hi = f16(s x), lo = f16(s x - hi) (round to nearest even, s a
// power-of-two scale so that typical magnitudes sit well inside the fp16 normal range; saturating at +-65504).  hi + lo
// carries 22 mantissa bits of s x; the product of two such operands is accumulated as hi*hi + hi*lo + lo*hi in fp32
// (the dropped lo*lo term is <= 2^-22 relative).
FP_DEVICE void split16_pack2(float a, float b, float scale, unsigned& hi, unsigned& lo) {
  a = __builtin_amdgcn_fmed3f(a * scale, -65504.f, 65504.f);
  b = __builtin_amdgcn_fmed3f(b * scale, -65504.f, 65504.f);
  const f16x2 h = __builtin_convertvector(f32x2{a, b}, f16x2);
  const f32x2 hf = __builtin_convertvector(h, f32x2);
  const f16x2 l = __builtin_convertvector(f32x2{a - hf[0], b - hf[1]}, f16x2);  // the residual is exact in fp32
  hi = __builtin_bit_cast(unsigned, h);
  lo = __builtin_bit_cast(unsigned, l);
}
// ... for values known to fit (|s x| <= 65504: the attention's probabilities, p <= 2 at scale 2^14): the same bits without the two clamps
FP_DEVICE void split16_pack2_inrange(float a, float b, float scale, unsigned& hi, unsigned& lo) {
  a *= scale;
  b *= scale;
  const f16x2 h = __builtin_convertvector(f32x2{a, b}, f16x2);
  const f32x2 hf = __builtin_convertvector(h, f32x2);
  const f16x2 l = __builtin_convertvector(f32x2{a - hf[0], b - hf[1]}, f16x2);
  hi = __builtin_bit_cast(unsigned, h);
  lo = __builtin_bit_cast(unsigned, l);
}
// ... and the running maximum of |s x| the caller reports at the end of the kernel (report_saturation)
FP_DEVICE void split16_pack2(float a, float b, float scale, unsigned& hi, unsigned& lo, float& amax) {
  amax = nanmax3(amax, fabsf(a * scale), fabsf(b * scale));  // v_maximum3_f32 with |.| modifiers
  split16_pack2(a, b, scale, hi, lo);
}
// position (in halves) of logical column c inside a split row; its lo half sits 32 halves further
FP_DEVICE int split16_pos(int c) { return ((c >> 5) << 6) + (c & 31); }

// ---- "f16f8" rows (the f16f8 mode: fp16 high halves, the two cross terms of a split product on the fp8 pipe).
// The same 4 bytes per logical element as a split-fp16 row, in groups of 64 columns (K % 64 == 0): group g = k / 64 holds
//   bytes [256 g,       256 g + 128)  hi = f16(s x)                      64 halves  -> hi * hi on v_mfma_f32_32x32x16_f16, as before
//   bytes [256 g + 128, 256 g + 192)  e4m3(hi 2^-7)                      64 bytes   \ the cross terms hi_a lo_w + lo_a hi_w as two
//   bytes [256 g + 192, 256 g + 256)  e4m3((s x - hi) 2^4)               64 bytes   / v_mfma_scale_f32_32x32x64_f8f6f4 (block scale 2^3)
// The e4m3 copies carry 4 significant bits of hi and of lo: a cross term is good to 2^-3 of 2^-11 of the product, so a product
// carries ~14 mantissa bits at the worst and the sum over K far more on average (measured on fc2, K = 4096: max error 1.3e-5 of the
// output scale, rms 2.5e-6 -- tools/sp_fp8cross.py) at 8 instead of 12 fp16-MFMA units per 64 k: 1.34x the three-fp16-MFMA form.
constexpr float FP_SX_HI_SCALE = 0.0078125f, FP_SX_LO_SCALE = 16.f;     // 2^-7 and 2^4: hi and lo (<= 16) into e4m3's range (<= 448)
// (hi 2^-7 <= 448 holds for |hi| <= 57344 only: between there and the fp16 limit 65504 the e4m3 copy of hi would clamp -- up to 12 % of that cross
//  term -- while the fp16 row itself is still exact.  Producers of f16f8 rows therefore report saturation from 57344 on: the head room of the mode is
//  +-3584 for LayerNorm / attention outputs and +-14336 for hidden activations, an eighth less than the split-fp16 rows'.)
constexpr float FP_SX_MAX = 57344.f;
constexpr unsigned FP_SX_MFMA_SCALE = 0x82828282u;                       // E8M0 127 + 3 in every byte: the block scale 2^3 that undoes 2^-7 x 2^4
FP_DEVICE int splitx_pos(int c) { return ((c >> 6) << 7) + (c & 63); }   // halves index of hi(column c), the row viewed as halves
FP_DEVICE int splitx_hi8(int c) { return ((c >> 6) << 8) + 128 + (c & 63); }  // byte offset of e4m3(hi) of column c; its e4m3(lo) sits 64 bytes further
// two columns -> hi = (f16, f16) and p8 = the bytes [e4m3(hi_a), e4m3(hi_b), e4m3(lo_a), e4m3(lo_b)]
FP_DEVICE void splitx_pack2(float a, float b, float scale, unsigned& hi, unsigned& p8) {
  a = __builtin_amdgcn_fmed3f(a * scale, -65504.f, 65504.f);
  b = __builtin_amdgcn_fmed3f(b * scale, -65504.f, 65504.f);
  const f16x2 h = __builtin_convertvector(f32x2{a, b}, f16x2);
  const f32x2 hf = __builtin_convertvector(h, f32x2);
  hi = __builtin_bit_cast(unsigned, h);
  int w = __builtin_amdgcn_cvt_pk_fp8_f32(clamp448(hf[0] * FP_SX_HI_SCALE), clamp448(hf[1] * FP_SX_HI_SCALE), 0, false);
  p8 = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(clamp448((a - hf[0]) * FP_SX_LO_SCALE), clamp448((b - hf[1]) * FP_SX_LO_SCALE), w, true);
}
FP_DEVICE void splitx_pack2(float a, float b, float scale, unsigned& hi, unsigned& p8, float& amax) {
  amax = nanmax3(amax, fabsf(a * scale), fabsf(b * scale));
  splitx_pack2(a, b, scale, hi, p8);
}
// stores of a lane's packed columns into a row image (`row` = first byte of the row, or of a tile's 64-column-aligned part of it):
// two adjacent columns c, c + 1 (c even) ...
FP_DEVICE void splitx_store2(char* row, int c, unsigned hi, unsigned p8) {
  *reinterpret_cast<unsigned*>(row + splitx_pos(c) * 2) = hi;
  *reinterpret_cast<unsigned short*>(row + splitx_hi8(c)) = (unsigned short)p8;
  *reinterpret_cast<unsigned short*>(row + splitx_hi8(c) + 64) = (unsigned short)(p8 >> 16);
}
// ... and four adjacent columns c .. c + 3 (c % 4 == 0) from two packed pairs
FP_DEVICE void splitx_store4(char* row, int c, unsigned hi01, unsigned p01, unsigned hi23, unsigned p23) {
  *reinterpret_cast<uint2*>(row + splitx_pos(c) * 2) = make_uint2(hi01, hi23);
  *reinterpret_cast<unsigned*>(row + splitx_hi8(c)) = __builtin_amdgcn_perm(p23, p01, 0x05040100u);        // hi8 of the four columns
  *reinterpret_cast<unsigned*>(row + splitx_hi8(c) + 64) = __builtin_amdgcn_perm(p23, p01, 0x07060302u);   // lo8 of the four columns
}

// ---- wave-level reductions (64 lanes)
FP_DEVICE float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
FP_DEVICE float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
FP_DEVICE unsigned long long wave_min_u64(unsigned long long v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    unsigned long long t = __shfl_xor(v, o, 64);
    v = t < v ? t : v;
  }
  return v;
}

// Non-negative floats order like their bit patterns: (d2, index) -> one u64 key whose
// unsigned order is "smaller distance first, ties -> lower index".
FP_DEVICE unsigned long long pack_dist_idx(float d2, unsigned idx) {
  return ((unsigned long long)__float_as_uint(d2) << 32) | idx;
}

// XCD-aware bijective block remap (guide section 5.5 T1): consecutive logical tiles share an XCD's L2.
FP_DEVICE unsigned xcd_remap(unsigned bid, unsigned nwg) {
  const unsigned nx = 8;
  unsigned q = nwg / nx, r = nwg % nx;
  unsigned xcd = bid % nx, idx = bid / nx;
  unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// ---- per-device launch-time caches.  hipFuncSetAttribute(MaxDynamicSharedMemorySize) and the CU count are properties
// of a (kernel, device) pair: the flag is kept per device id, so a process that drives several GPUs sets the attribute
// on each of them.  The cached values are idempotent (two host threads racing on a flag both set the same attribute),
// which is the only mutable state the library keeps.
constexpr int FP_MAX_DEVICES = 64;
struct FpDeviceOnce {
  std::atomic<unsigned char> done[FP_MAX_DEVICES];
};
// true when `o` has not been marked on the calling thread's current device yet (always true for device ids past the table)
static inline bool fp_first_on_device(FpDeviceOnce& o) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= FP_MAX_DEVICES) return true;
  if (o.done[dev].load(std::memory_order_acquire)) return false;
  o.done[dev].store(1, std::memory_order_release);
  return true;
}
// raise a kernel's dynamic-LDS limit once per device
template <class K>
static inline void fp_allow_dynamic_lds(FpDeviceOnce& o, K kernel, int bytes) {
  if (fp_first_on_device(o)) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}
// compute units of the current device (cached per device)
static inline int fp_num_cus() {
  static std::atomic<int> cus[FP_MAX_DEVICES];
  int dev = 0, n = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= FP_MAX_DEVICES) return 256;
  n = cus[dev].load(std::memory_order_relaxed);
  if (n > 0) return n;
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
  cus[dev].store(n, std::memory_order_relaxed);
  return n;
}
