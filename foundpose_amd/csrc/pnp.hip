// Batched PnP-RANSAC + Levenberg-Marquardt refinement: the step behind the matching path.
//
// Reference behaviour restated (paths under /root/reference):
//   utils/pnp_util.py:20-84 ... estimate_pose: cv2.solvePnPRansac(objectPoints, imagePoints, K, iterationsCount,
//                               reprojectionError, confidence, flags=SOLVEPNP_ITERATIVE) + optional cv2.solvePnPRefineLM on
//                               the inliers; quality = number of RANSAC inliers
//   scripts/infer.py:552-602 .. one call per retrieved template (>= 6 correspondences), best quality wins (host side)
// cv2 (opencv-python 4.5.5.62) is not in the image, so its arithmetic cannot be pinned; this is its published scheme with
// a GPU-shaped minimal solver: RANSAC over minimal samples, inliers = reprojection error <= threshold, the best model is
// the first one with the highest inlier count within the adaptively shortened iteration budget
// (RANSACUpdateNumIters), then iterative refinement of that model on its inliers.  Differences, stated:
//   * minimal solver: P3P (Grunert's quartic, 3 points) + a 4th point to pick among its <= 4 solutions -- what OpenCV's
//     RANSAC uses for the P3P flags -- instead of EPnP on 5 points; hypotheses are independent, so all of them are
//     generated and scored in parallel and the adaptive stop is replayed over their inlier counts afterwards;
//   * sampling: a counter-based hash of (seed, pair, hypothesis) instead of cv::RNG;
//   * refinement: one Levenberg-Marquardt loop (<= 20 iterations for the solvePnP stage + <= 20 for solvePnPRefineLM) on
//     the reprojection error in pixels, rotation updated on the manifold.
// One 256-thread workgroup per (detection, template slot) pair; everything in fp64.  HBM traffic is a few KB per pair:
// the kernel is latency / fp64-VALU bound.
#include "common.hpp"
#include "kernels.hpp"

namespace {

constexpr int PNP_THREADS = 256;
constexpr int PNP_MAX_ITERS = 4096;

struct Pose {
  double R[9];
  double t[3];
};

FP_DEVICE unsigned long long mix64(unsigned long long z) {  // splitmix64 finaliser
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

FP_DEVICE void cross3(const double* a, const double* b, double* c) {
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
}
FP_DEVICE double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
FP_DEVICE bool normalize3(double* a) {
  const double n = sqrt(dot3(a, a));
  if (!(n > 1e-300)) return false;
  a[0] /= n; a[1] /= n; a[2] /= n;
  return true;
}

// orthonormal frame of a triangle: e1 along Q1-Q0, e3 normal, e2 = e3 x e1 (columns of F)
FP_DEVICE bool tri_frame(const double* q0, const double* q1, const double* q2, double* F) {
  double e1[3] = {q1[0] - q0[0], q1[1] - q0[1], q1[2] - q0[2]};
  double d2[3] = {q2[0] - q0[0], q2[1] - q0[1], q2[2] - q0[2]};
  double e3[3], e2[3];
  if (!normalize3(e1)) return false;
  cross3(e1, d2, e3);
  if (!normalize3(e3)) return false;
  cross3(e3, e1, e2);
  for (int r = 0; r < 3; ++r) { F[r * 3 + 0] = e1[r]; F[r * 3 + 1] = e2[r]; F[r * 3 + 2] = e3[r]; }
  return true;
}

// Real roots of x^4 + b3 x^3 + b2 x^2 + b1 x + b0 (Durand-Kerner on the four complex roots, then two Newton steps).
FP_DEVICE int quartic_real_roots(double b3, double b2, double b1, double b0, double* out) {
  double re[4], im[4];
  const double bound = 1.0 + fmax(fmax(fabs(b3), fabs(b2)), fmax(fabs(b1), fabs(b0)));
  const double r0 = fmin(bound, 1e3);
  {  // starting points on a circle, off the real axis
    double cr = 0.4 * r0, ci = 0.9 * r0;
    for (int k = 0; k < 4; ++k) {
      re[k] = cr; im[k] = ci;
      const double nr = cr * 0.4 - ci * 0.9, ni = cr * 0.9 + ci * 0.4;
      cr = nr; ci = ni;
    }
  }
  for (int it = 0; it < 60; ++it) {
    double moved = 0.0;
    for (int k = 0; k < 4; ++k) {
      // p(z) by Horner, complex
      double pr = 1.0, pi = 0.0;
      const double c[4] = {b3, b2, b1, b0};
      for (int j = 0; j < 4; ++j) {
        const double nr = pr * re[k] - pi * im[k] + c[j], ni = pr * im[k] + pi * re[k];
        pr = nr; pi = ni;
      }
      double dr = 1.0, di = 0.0;
      for (int j = 0; j < 4; ++j) {
        if (j == k) continue;
        const double ar = re[k] - re[j], ai = im[k] - im[j];
        const double nr = dr * ar - di * ai, ni = dr * ai + di * ar;
        dr = nr; di = ni;
      }
      const double den = dr * dr + di * di;
      if (den < 1e-300) continue;
      const double qr = (pr * dr + pi * di) / den, qi = (pi * dr - pr * di) / den;
      re[k] -= qr; im[k] -= qi;
      moved = fmax(moved, fabs(qr) + fabs(qi));
    }
    if (moved < 1e-15 * r0) break;
  }
  int n = 0;
  for (int k = 0; k < 4; ++k) {
    if (fabs(im[k]) > 1e-6 * (1.0 + fabs(re[k]))) continue;
    double x = re[k];
    for (int s = 0; s < 2; ++s) {
      const double p = (((x + b3) * x + b2) * x + b1) * x + b0;
      const double d = ((4.0 * x + 3.0 * b3) * x + 2.0 * b2) * x + b1;
      if (fabs(d) > 1e-300) x -= p / d;
    }
    out[n++] = x;
  }
  return n;
}

// P3P, Grunert's formulation: X world points, f unit bearings -> up to 4 poses (x_cam = R X + t).
FP_DEVICE int p3p_grunert(const double X[3][3], const double f[3][3], Pose* sols) {
  double d12[3], d02[3], d01[3];
  for (int r = 0; r < 3; ++r) { d12[r] = X[1][r] - X[2][r]; d02[r] = X[0][r] - X[2][r]; d01[r] = X[0][r] - X[1][r]; }
  const double a2 = dot3(d12, d12), b2 = dot3(d02, d02), c2 = dot3(d01, d01);
  if (!(a2 > 1e-18 && b2 > 1e-18 && c2 > 1e-18)) return 0;
  const double ca = dot3(f[1], f[2]), cb = dot3(f[0], f[2]), cg = dot3(f[0], f[1]);
  const double q = (a2 - c2) / b2, p = (a2 + c2) / b2;
  const double A4 = (q - 1.0) * (q - 1.0) - 4.0 * c2 / b2 * ca * ca;
  const double A3 = 4.0 * (q * (1.0 - q) * cb - (1.0 - p) * ca * cg + 2.0 * c2 / b2 * ca * ca * cb);
  const double A2 = 2.0 * (q * q - 1.0 + 2.0 * q * q * cb * cb + 2.0 * (b2 - c2) / b2 * ca * ca - 4.0 * p * ca * cb * cg + 2.0 * (b2 - a2) / b2 * cg * cg);
  const double A1 = 4.0 * (-q * (1.0 + q) * cb + 2.0 * a2 / b2 * cg * cg * cb - (1.0 - p) * ca * cg);
  const double A0 = (1.0 + q) * (1.0 + q) - 4.0 * a2 / b2 * cg * cg;
  if (!(fabs(A4) > 1e-12 * (fabs(A3) + fabs(A2) + fabs(A1) + fabs(A0) + 1e-300))) return 0;
  double roots[4];
  const int nr = quartic_real_roots(A3 / A4, A2 / A4, A1 / A4, A0 / A4, roots);
  double Fw[9];
  if (!tri_frame(X[0], X[1], X[2], Fw)) return 0;
  int n = 0;
  for (int k = 0; k < nr; ++k) {
    const double v = roots[k];
    if (!(v > 0.0)) continue;
    const double den = 2.0 * (cg - v * ca);
    if (!(fabs(den) > 1e-12)) continue;
    const double u = ((-1.0 + q) * v * v - 2.0 * q * cb * v + 1.0 + q) / den;
    if (!(u > 0.0)) continue;
    const double s1sq = c2 / (1.0 + u * u - 2.0 * u * cg);
    if (!(s1sq > 0.0)) continue;
    const double s1 = sqrt(s1sq), s2 = u * s1, s3 = v * s1;
    double P[3][3];
    for (int r = 0; r < 3; ++r) { P[0][r] = s1 * f[0][r]; P[1][r] = s2 * f[1][r]; P[2][r] = s3 * f[2][r]; }
    double Fc[9];
    if (!tri_frame(P[0], P[1], P[2], Fc)) continue;
    Pose& o = sols[n];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) o.R[i * 3 + j] = Fc[i * 3 + 0] * Fw[j * 3 + 0] + Fc[i * 3 + 1] * Fw[j * 3 + 1] + Fc[i * 3 + 2] * Fw[j * 3 + 2];  // Fc Fw^T
    for (int i = 0; i < 3; ++i) o.t[i] = P[0][i] - (o.R[i * 3 + 0] * X[0][0] + o.R[i * 3 + 1] * X[0][1] + o.R[i * 3 + 2] * X[0][2]);
    ++n;
  }
  return n;
}

// squared reprojection error in pixels; 1e300 behind the camera
FP_DEVICE double reproj_err2(const Pose& P, const float* X, const float* uv, double fx, double fy, double cx, double cy) {
  const double x = P.R[0] * X[0] + P.R[1] * X[1] + P.R[2] * X[2] + P.t[0];
  const double y = P.R[3] * X[0] + P.R[4] * X[1] + P.R[5] * X[2] + P.t[1];
  const double z = P.R[6] * X[0] + P.R[7] * X[1] + P.R[8] * X[2] + P.t[2];
  if (!(z > 1e-9)) return 1e300;
  const double du = fx * x / z + cx - uv[0], dv = fy * y / z + cy - uv[1];
  return du * du + dv * dv;
}

// hypothesis h of a pair: sample 4 distinct correspondences, P3P on the first three, the fourth picks the solution
FP_DEVICE bool hypothesis(unsigned long long key, int N, const float* X3, const float* UV, double fx, double fy, double cx, double cy, Pose* out) {
  int id[4];
  unsigned long long s = key;
  for (int j = 0; j < 4; ++j) {
    for (int attempt = 0;; ++attempt) {
      s = mix64(s);
      const int c = (int)(s % (unsigned long long)N);
      bool dup = false;
      for (int i = 0; i < j; ++i) dup |= id[i] == c;
      if (!dup) { id[j] = c; break; }
      if (attempt > 64) return false;
    }
  }
  double X[3][3], f[3][3];
  for (int j = 0; j < 3; ++j) {
    for (int r = 0; r < 3; ++r) X[j][r] = X3[id[j] * 3 + r];
    f[j][0] = (UV[id[j] * 2 + 0] - cx) / fx;
    f[j][1] = (UV[id[j] * 2 + 1] - cy) / fy;
    f[j][2] = 1.0;
    normalize3(f[j]);
  }
  Pose sols[4];
  const int n = p3p_grunert(X, f, sols);
  double best = 1e299;
  int bi = -1;
  for (int k = 0; k < n; ++k) {
    const double e = reproj_err2(sols[k], X3 + id[3] * 3, UV + id[3] * 2, fx, fy, cx, cy);
    if (e < best) { best = e; bi = k; }
  }
  if (bi < 0) return false;
  *out = sols[bi];
  return true;
}

FP_DEVICE double block_sum(double v, double* red, int tid) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// exp of a rotation vector times R (left perturbation)
FP_DEVICE void rot_update(const double* w, const double* R, double* Rn) {
  const double th2 = dot3(w, w), th = sqrt(th2);
  double a, b;  // exp([w]x) = I + a [w]x + b [w]x^2
  if (th < 1e-8) { a = 1.0 - th2 / 6.0; b = 0.5 - th2 / 24.0; }
  else { a = sin(th) / th; b = (1.0 - cos(th)) / th2; }
  const double K[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
  double K2[9], E[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) K2[i * 3 + j] = K[i * 3 + 0] * K[0 * 3 + j] + K[i * 3 + 1] * K[1 * 3 + j] + K[i * 3 + 2] * K[2 * 3 + j];
  for (int i = 0; i < 9; ++i) E[i] = (i % 4 == 0 ? 1.0 : 0.0) + a * K[i] + b * K2[i];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) Rn[i * 3 + j] = E[i * 3 + 0] * R[0 * 3 + j] + E[i * 3 + 1] * R[1 * 3 + j] + E[i * 3 + 2] * R[2 * 3 + j];
}

// RANSACUpdateNumIters of OpenCV's point-set registrator
FP_DEVICE int update_num_iters(double p, double ep, int model_points, int max_iters) {
  p = fmin(fmax(p, 0.0), 1.0);
  ep = fmin(fmax(ep, 0.0), 1.0);
  double num = fmax(1.0 - p, 2.2250738585072014e-308);
  double denom = 1.0 - pow(1.0 - ep, (double)model_points);
  if (denom < 2.2250738585072014e-308) return 0;
  num = log(num);
  denom = log(denom);
  return (denom >= 0 || -num >= max_iters * (-denom)) ? max_iters : (int)rint(num / denom);
}

__global__ __launch_bounds__(PNP_THREADS) void pnp_ransac_kernel(PnpArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* X3 = reinterpret_cast<float*>(smem);                 // [k_max, 3]
  float* UV = X3 + (size_t)a.k_max * 3;                       // [k_max, 2]
  int* cnt = reinterpret_cast<int*>(UV + (size_t)a.k_max * 2);  // [iters]
  unsigned char* inl = reinterpret_cast<unsigned char*>(cnt + a.iters);  // [k_max]
  __shared__ double red[4];
  __shared__ double acc[4][28];
  __shared__ Pose cur, cand;
  __shared__ int s_best, s_count, s_flag;
  __shared__ double s_lambda, s_cost;
  __shared__ double s_A[21], s_g[6];

  const int pair = blockIdx.x, tid = threadIdx.x;
  const int det = pair / a.n_slots;
  const int N = min(a.counts[pair], a.k_max);
  const double fx = a.cam[det * 4 + 0], fy = a.cam[det * 4 + 1], cx = a.cam[det * 4 + 2], cy = a.cam[det * 4 + 3];
  auto fail = [&]() {
    if (tid == 0) {
      a.success[pair] = 0;
      a.n_inliers[pair] = 0;
      for (int i = 0; i < 9; ++i) a.R[(size_t)pair * 9 + i] = (i % 4 == 0) ? 1.0 : 0.0;
      for (int i = 0; i < 3; ++i) a.t[(size_t)pair * 3 + i] = 0.0;
      if (a.ransac_pose) for (int i = 0; i < 12; ++i) a.ransac_pose[(size_t)pair * 12 + i] = 0.0;
    }
    for (int i = tid; i < a.k_max; i += PNP_THREADS) a.inlier_mask[(size_t)pair * a.k_max + i] = 0;
  };
  if (N < a.min_corresp) { fail(); return; }  // block-uniform
  for (int i = tid; i < N * 3; i += PNP_THREADS) X3[i] = a.coord_3d[(size_t)pair * a.k_max * 3 + i];
  for (int i = tid; i < N * 2; i += PNP_THREADS) UV[i] = a.coord_2d[(size_t)pair * a.k_max * 2 + i];
  __syncthreads();
  const double thr2 = a.thresh * a.thresh;
  const unsigned long long base = mix64(a.seed ^ ((unsigned long long)pair * 0xD6E8FEB86659FD93ull));

  // ---- all hypotheses, scored in parallel
  for (int h = tid; h < a.iters; h += PNP_THREADS) {
    Pose P;
    int c = 0;
    if (hypothesis(base + (unsigned long long)h * 0x9E3779B97F4A7C15ull, N, X3, UV, fx, fy, cx, cy, &P))
      for (int p = 0; p < N; ++p) c += reproj_err2(P, X3 + p * 3, UV + p * 2, fx, fy, cx, cy) <= thr2 ? 1 : 0;
    cnt[h] = c;
  }
  __syncthreads();
  // ---- the sequential best-model rule with the adaptive iteration budget, replayed over the counts
  if (tid == 0) {
    int best = -1, best_c = 3, niters = a.iters;  // a model must beat model_points - 1 = 3 inliers
    for (int h = 0; h < niters; ++h) {
      if (cnt[h] > best_c) {
        best_c = cnt[h];
        best = h;
        niters = update_num_iters(a.conf, (double)(N - best_c) / N, 4, niters);
      }
    }
    s_best = best;
    s_count = best_c;
  }
  __syncthreads();
  if (s_best < 0) { fail(); return; }
  if (tid == (s_best % PNP_THREADS)) {  // the owner regenerates the winning hypothesis (deterministic)
    Pose P;
    hypothesis(base + (unsigned long long)s_best * 0x9E3779B97F4A7C15ull, N, X3, UV, fx, fy, cx, cy, &P);
    cur = P;
  }
  __syncthreads();
  for (int p = tid; p < a.k_max; p += PNP_THREADS) {
    const unsigned char m = p < N && reproj_err2(cur, X3 + p * 3, UV + p * 2, fx, fy, cx, cy) <= thr2;
    if (p < N) inl[p] = m;
    a.inlier_mask[(size_t)pair * a.k_max + p] = m;
  }
  if (tid == 0 && a.ransac_pose) {
    for (int i = 0; i < 9; ++i) a.ransac_pose[(size_t)pair * 12 + i] = cur.R[i];
    for (int i = 0; i < 3; ++i) a.ransac_pose[(size_t)pair * 12 + 9 + i] = cur.t[i];
  }
  __syncthreads();

  // ---- Levenberg-Marquardt on the inliers: cost, J^T J (21 upper entries) and J^T r (6) of a pose, block-reduced
  auto normal_eq = [&](const Pose& P, bool want_jac) -> double {
    double v[28];
#pragma unroll
    for (int i = 0; i < 28; ++i) v[i] = 0.0;
    for (int p = tid; p < N; p += PNP_THREADS) {
      if (!inl[p]) continue;
      const float* Xp = X3 + p * 3;
      const double x = P.R[0] * Xp[0] + P.R[1] * Xp[1] + P.R[2] * Xp[2] + P.t[0];
      const double y = P.R[3] * Xp[0] + P.R[4] * Xp[1] + P.R[5] * Xp[2] + P.t[1];
      const double z = P.R[6] * Xp[0] + P.R[7] * Xp[1] + P.R[8] * Xp[2] + P.t[2];
      if (!(z > 1e-9)) {  // an inlier behind (or on) the camera plane: like reproj_err2, a cost no step is accepted with
        v[27] += 1e30;
        continue;
      }
      const double iz = 1.0 / z;
      const double ru = fx * x * iz + cx - UV[p * 2 + 0], rv = fy * y * iz + cy - UV[p * 2 + 1];
      v[27] += ru * ru + rv * rv;
      if (!want_jac) continue;
      // d(u,v)/d(xc), then d(xc)/d(w, dt) for R' = exp([w]x) R, t' = t + dt:  d xc / dw = -[R X]x, d xc / d dt = I
      const double ux = fx * iz, uz = -fx * x * iz * iz, vy = fy * iz, vz = -fy * y * iz * iz;
      const double px = x - P.t[0], py = y - P.t[1], pz = z - P.t[2];
      const double Ju[6] = {uz * py, ux * pz - uz * px, -ux * py, ux, 0.0, uz};       // row of u: [d/dw (3), d/dt (3)]
      const double Jv[6] = {-vy * pz + vz * py, -vz * px, vy * px, 0.0, vy, vz};
      int k = 0;
#pragma unroll
      for (int i = 0; i < 6; ++i) {
#pragma unroll
        for (int j = i; j < 6; ++j) v[k++] += Ju[i] * Ju[j] + Jv[i] * Jv[j];
      }
#pragma unroll
      for (int i = 0; i < 6; ++i) v[21 + i] += Ju[i] * ru + Jv[i] * rv;
    }
    const int first = want_jac ? 0 : 27;
    for (int i = first; i < 28; ++i) {
      double s = v[i];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
      if ((tid & 63) == 0) acc[tid >> 6][i] = s;
    }
    __syncthreads();
    if (tid == 0 && want_jac) {
      for (int i = 0; i < 21; ++i) s_A[i] = acc[0][i] + acc[1][i] + acc[2][i] + acc[3][i];
      for (int i = 0; i < 6; ++i) s_g[i] = acc[0][21 + i] + acc[1][21 + i] + acc[2][21 + i] + acc[3][21 + i];
    }
    const double cost = acc[0][27] + acc[1][27] + acc[2][27] + acc[3][27];
    __syncthreads();
    return cost;
  };

  double cost = normal_eq(cur, true);
  if (tid == 0) { s_lambda = 1e-3; s_cost = cost; }
  __syncthreads();
  for (int it = 0; it < a.lm_iters; ++it) {
    if (tid == 0) {
      // (A + lambda diag(A)) d = -g by Cholesky
      double M[6][6], L[6][6], rhs[6], d[6];
      int k = 0;
      for (int i = 0; i < 6; ++i)
        for (int j = i; j < 6; ++j) { M[i][j] = M[j][i] = s_A[k++]; }
      for (int i = 0; i < 6; ++i) { M[i][i] += s_lambda * fmax(M[i][i], 1e-12); rhs[i] = -s_g[i]; }
      bool ok = true;
      for (int i = 0; i < 6 && ok; ++i)
        for (int j = 0; j <= i; ++j) {
          double s = M[i][j];
          for (int q = 0; q < j; ++q) s -= L[i][q] * L[j][q];
          if (i == j) { if (!(s > 0.0)) { ok = false; break; } L[i][i] = sqrt(s); }
          else L[i][j] = s / L[j][j];
        }
      if (ok) {
        double y[6];
        for (int i = 0; i < 6; ++i) { double s = rhs[i]; for (int q = 0; q < i; ++q) s -= L[i][q] * y[q]; y[i] = s / L[i][i]; }
        for (int i = 5; i >= 0; --i) { double s = y[i]; for (int q = i + 1; q < 6; ++q) s -= L[q][i] * d[q]; d[i] = s / L[i][i]; }
        rot_update(d, cur.R, cand.R);
        for (int i = 0; i < 3; ++i) cand.t[i] = cur.t[i] + d[3 + i];
        const double step = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]) + sqrt(d[3] * d[3] + d[4] * d[4] + d[5] * d[5]) / (1.0 + sqrt(dot3(cur.t, cur.t)));
        s_flag = step < 1e-14 ? 2 : 1;  // 2: converged
      } else {
        s_flag = 0;  // not positive definite at this damping: raise it
      }
    }
    __syncthreads();
    const int flag = s_flag;
    __syncthreads();  // every wave has read the solve status before thread 0 may write s_flag again (below / next iteration)
    if (flag == 2) break;
    if (flag == 0) {
      if (tid == 0) s_lambda *= 10.0;
      __syncthreads();
      const bool give_up = s_lambda > 1e12;
      __syncthreads();  // ... and s_lambda before the next iteration's solve touches it
      if (give_up) break;
      continue;
    }
    // cost + normal equations at the candidate; kept only if the step is accepted
    __shared__ double sA_keep[21], sg_keep[6];
    if (tid == 0) { for (int i = 0; i < 21; ++i) sA_keep[i] = s_A[i]; for (int i = 0; i < 6; ++i) sg_keep[i] = s_g[i]; }
    __syncthreads();
    const double c_new = normal_eq(cand, true);
    if (tid == 0) {
      if (c_new < s_cost) {
        const double rel = (s_cost - c_new) / fmax(s_cost, 1e-300);
        cur = cand;
        s_cost = c_new;
        s_lambda = fmax(s_lambda * 0.1, 1e-15);
        s_flag = rel < 1e-16 ? 2 : 1;
      } else {
        for (int i = 0; i < 21; ++i) s_A[i] = sA_keep[i];
        for (int i = 0; i < 6; ++i) s_g[i] = sg_keep[i];
        s_lambda *= 10.0;
        s_flag = s_lambda > 1e12 ? 2 : 1;
      }
    }
    __syncthreads();
    const int step_flag = s_flag;  // read into a register, then a barrier: the next iteration's thread 0 rewrites s_flag
    __syncthreads();
    if (step_flag == 2) break;
  }
  __syncthreads();
  if (tid == 0) {
    a.success[pair] = 1;
    a.n_inliers[pair] = s_count;
    for (int i = 0; i < 9; ++i) a.R[(size_t)pair * 9 + i] = cur.R[i];
    for (int i = 0; i < 3; ++i) a.t[(size_t)pair * 3 + i] = cur.t[i];
  }
}

}  // namespace

int launch_pnp_ransac(const PnpArgs& a, int num_pairs, hipStream_t st) {
  FP_REQUIRE(a.k_max >= 4 && a.k_max <= 4096, "pnp_ransac: k_max must be in [4, 4096] (got %d)", a.k_max);
  FP_REQUIRE(a.iters >= 1 && a.iters <= PNP_MAX_ITERS, "pnp_ransac: iterations must be in [1, %d] (got %d)", PNP_MAX_ITERS, a.iters);
  FP_REQUIRE(a.thresh > 0.0 && a.conf > 0.0 && a.conf <= 1.0, "pnp_ransac: bad threshold / confidence");
  FP_REQUIRE(a.lm_iters >= 0 && a.n_slots >= 1 && a.min_corresp >= 4, "pnp_ransac: bad lm_iters / n_slots / min_corresp");
  if (num_pairs == 0) return FP_OK;
  const size_t lds = (size_t)a.k_max * 5 * 4 + (size_t)a.iters * 4 + (size_t)a.k_max;
  static FpDeviceOnce attr;
  fp_allow_dynamic_lds(attr, &pnp_ransac_kernel, 4096 * 5 * 4 + PNP_MAX_ITERS * 4 + 4096);
  hipLaunchKernelGGL(pnp_ransac_kernel, dim3(num_pairs), dim3(PNP_THREADS), lds, st, a);
  FP_CHECK_LAUNCH("pnp_ransac");
  return FP_OK;
}
