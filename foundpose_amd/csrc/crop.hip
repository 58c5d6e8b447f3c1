// Crop producer: warps detections from the input image into the square viewport of a virtual crop camera.
//
// Replaces, per detection, utils/misc.py:458-519 `warp_image` (destination pixel -> eye ray -> world -> source eye ->
// source window, fp64 numpy, cast to fp32 maps, then cv2.remap) as scripts/infer.py:433-450 calls it: bilinear for the
// RGB image (INTER_AREA falls back to INTER_LINEAR inside cv2.remap), nearest for the modal mask, constant border 0.
// One thread per destination pixel; the whole batch of crops is one launch and the RGB result is written channel-major
// ([B,3,S,S], what infer.py:466-468 builds with array_to_tensor().permute(2,0,1)) -- the layout patchify reads.
//
// Arithmetic is kept operation-for-operation like the reference chain: every product / sum / quotient / sqrt below is
// an individually rounded fp64 operation (no contraction) in the order numpy evaluates it elementwise; the matrix
// products are k-ordered mul-add chains (numpy hands them to BLAS, whose last-bit behaviour is not specified --
// fixtures taken from the reference agree except for isolated 1-ulp pixels of the fp32 map, tests/test_crop_cpu.py).
// cv2.remap itself is restated from OpenCV 4.5 imgwarp.cpp semantics (cv2 is absent from the image: unpinned):
//   fixed point  sx = cvRound(map_x * 32), sy likewise (round half to even); ix = sx >> 5, ax = sx & 31
//   bilinear     ((S00*w00 + S01*w01) + S10*w10) + S11*w11 in fp32, w = (1-ay/32 | ay/32) x (1-ax/32 | ax/32) (exact),
//                taps outside the image read the border value 0
//   nearest      S[cvRound(map_y)][cvRound(map_x)], 0 outside
#include "common.hpp"
#include "kernels.hpp"
#include "../../include/foundpose_amd.h"

namespace {

FP_DEVICE double dmul(double a, double b) { return __dmul_rn(a, b); }
FP_DEVICE double dadd(double a, double b) { return __dadd_rn(a, b); }
FP_DEVICE double dot3(double a0, double a1, double a2, double b0, double b1, double b2) {
  return dadd(dadd(dmul(a0, b0), dmul(a1, b1)), dmul(a2, b2));
}

__global__ __launch_bounds__(256) void warp_crops_kernel(WarpArgs a) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6), b = blockIdx.z;
  if (x >= a.out_w || y >= a.out_h) return;
  const double* p = a.params + (size_t)b * 32;  // dst f[2] c[2] R[9] t[3] | src f[2] c[2] R[9] t[3]
  // window_to_eye of the crop camera: q = (w - c) / f, v = normalized([qx, qy, 1])
  const double qx = __ddiv_rn(dadd((double)x, -p[2]), p[0]), qy = __ddiv_rn(dadd((double)y, -p[3]), p[1]);
  double n = __dsqrt_rn(dadd(dadd(dmul(qx, qx), dmul(qy, qy)), 1.0));
  n = fmax(5.43e-20, n);
  const double vx = __ddiv_rn(qx, n), vy = __ddiv_rn(qy, n), vz = __ddiv_rn(1.0, n);
  // eye_to_world: v @ R_d^T + t_d
  const double* Rd = p + 4;
  const double wx = dadd(dot3(vx, vy, vz, Rd[0], Rd[1], Rd[2]), p[13]);
  const double wy = dadd(dot3(vx, vy, vz, Rd[3], Rd[4], Rd[5]), p[14]);
  const double wz = dadd(dot3(vx, vy, vz, Rd[6], Rd[7], Rd[8]), p[15]);
  // world_to_eye of the source camera: (w - t_s) @ R_s
  const double* q = p + 16;
  const double* Rs = q + 4;
  const double dx = dadd(wx, -q[13]), dy = dadd(wy, -q[14]), dz = dadd(wz, -q[15]);
  const double ex = dot3(dx, dy, dz, Rs[0], Rs[3], Rs[6]);
  const double ey = dot3(dx, dy, dz, Rs[1], Rs[4], Rs[7]);
  const double ez = dot3(dx, dy, dz, Rs[2], Rs[5], Rs[8]);
  // eye_to_window: (e.xy / e.z) * f + c; points behind the source camera map to -1
  double mx = dadd(dmul(__ddiv_rn(ex, ez), q[0]), q[2]), my = dadd(dmul(__ddiv_rn(ey, ez), q[1]), q[3]);
  if (a.depth_check && ez < 0.0) mx = my = -1.0;
  const float fx = (float)mx, fy = (float)my;  // .astype(np.float32)
  const size_t plane = (size_t)a.out_h * a.out_w, pix = (size_t)y * a.out_w + x;
  if (a.map_out) {
    a.map_out[((size_t)b * 2 + 0) * plane + pix] = fx;
    a.map_out[((size_t)b * 2 + 1) * plane + pix] = fy;
  }
  const int img = a.src_index ? a.src_index[b] : b;
  if (a.mode == FP_WARP_NEAREST) {
    const unsigned char* src = reinterpret_cast<const unsigned char*>(a.src) + (size_t)img * a.src_h * a.src_w;
    const int sx = __float2int_rn(fx), sy = __float2int_rn(fy);
    const bool in = sx >= 0 && sx < a.src_w && sy >= 0 && sy < a.src_h;
    reinterpret_cast<unsigned char*>(a.out)[(size_t)b * plane + pix] = in ? src[(size_t)sy * a.src_w + sx] : 0;
    return;
  }
  const float* src = reinterpret_cast<const float*>(a.src) + (size_t)img * a.src_h * a.src_w * a.channels;
  const int sx = __float2int_rn(__fmul_rn(fx, 32.f)), sy = __float2int_rn(__fmul_rn(fy, 32.f));
  const int ix = sx >> 5, iy = sy >> 5;
  const float ax = (float)(sx & 31) * 0.03125f, ay = (float)(sy & 31) * 0.03125f;
  const float w00 = (1.f - ay) * (1.f - ax), w01 = (1.f - ay) * ax, w10 = ay * (1.f - ax), w11 = ay * ax;  // exact
  const bool x0 = ix >= 0 && ix < a.src_w, x1 = ix + 1 >= 0 && ix + 1 < a.src_w;
  const bool y0 = iy >= 0 && iy < a.src_h, y1 = iy + 1 >= 0 && iy + 1 < a.src_h;
  const float* r0 = src + ((size_t)iy * a.src_w + ix) * a.channels;
  const float* r1 = r0 + (size_t)a.src_w * a.channels;
  float* out = reinterpret_cast<float*>(a.out) + (size_t)b * a.channels * plane + pix;
  for (int ch = 0; ch < a.channels; ++ch) {
    const float s00 = y0 && x0 ? r0[ch] : 0.f, s01 = y0 && x1 ? r0[a.channels + ch] : 0.f;
    const float s10 = y1 && x0 ? r1[ch] : 0.f, s11 = y1 && x1 ? r1[a.channels + ch] : 0.f;
    const float v = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(s00, w00), __fmul_rn(s01, w01)), __fmul_rn(s10, w10)), __fmul_rn(s11, w11));
    out[(size_t)ch * plane] = v;
  }
}

}  // namespace

int launch_warp_crops(const WarpArgs& a, hipStream_t st) {
  dim3 grid(cdiv(a.out_w, 64), cdiv(a.out_h, 4), a.batch);
  hipLaunchKernelGGL(warp_crops_kernel, grid, dim3(256), 0, st, a);
  FP_CHECK_LAUNCH("warp_crops");
  return FP_OK;
}
