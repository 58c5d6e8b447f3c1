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

// ---- split-fp16 operands of the f16x3 mode (near-exact fp32 products on the fp16 MFMA).
// A logical fp32 row x[0..K) (K % 32 == 0) is stored as 2K halves: group g = k / 32 holds hi(x[32g .. 32g+31]) in halves
// [64g, 64g + 32) and lo(...) in [64g + 32, 64g + 64), with hi = f16(s x), lo = f16(s x - hi) (round to nearest even, s a
// power-of-two scale so that typical magnitudes sit well inside the fp16 normal range; saturating at +-65504).  hi + lo
// carries 22 mantissa bits of s x; the product of two such operands is accumulated as hi*hi + hi*lo + lo*hi in fp32
// (the dropped lo*lo term is <= 2^-22 relative).
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
FP_DEVICE void split16_pack2(float a, float b, float scale, unsigned& hi, unsigned& lo) {
  a = __builtin_amdgcn_fmed3f(a * scale, -65504.f, 65504.f);
  b = __builtin_amdgcn_fmed3f(b * scale, -65504.f, 65504.f);
  const f16x2 h = __builtin_convertvector(f32x2{a, b}, f16x2);
  const f32x2 hf = __builtin_convertvector(h, f32x2);
  const f16x2 l = __builtin_convertvector(f32x2{a - hf[0], b - hf[1]}, f16x2);  // the residual is exact in fp32
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
