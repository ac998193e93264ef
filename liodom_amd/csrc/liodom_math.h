// liodom_math.h — arithmetic leaf functions of the MI355X path (host + device).
//
// Everything here is straight-line FP code without wave intrinsics, so the same source is used
// by the HIP kernels and can be compiled by g++ for CPU cross-checks against the oracle
// (tests/hostcheck.cc).  Build with -ffp-contract=off: edge sets and kNN ordering must be
// bit-exact with the reference's non-FMA x86-64 arithmetic.
//
// Reference citations are relative to /root/reference.
#pragma once
#include <cfloat>
#include <cmath>
#include <cstdint>

#if defined(__HIPCC__)
#define LD_HD __host__ __device__ __forceinline__
#define LD_UNROLL _Pragma("unroll")
// The LM evaluation is not on the bit-exact path (pose tolerance 1e-4; the oracle differentiates
// a different expression tree anyway), so FMA contraction is allowed there: it halves the FP64
// instruction count of the dominant loop.  Everything else in this header is compiled with
// -ffp-contract=off.
#define LD_FP_CONTRACT_FAST _Pragma("clang fp contract(fast)")
#else
#define LD_HD inline
#define LD_UNROLL
#define LD_FP_CONTRACT_FAST
#endif
// reciprocal square root: one v_rsq_f64 + refinement on the device instead of a square root
// followed by a division (each ~25 instructions in FP64)
#if defined(__HIP_DEVICE_COMPILE__)
#define LD_RSQRT(x) rsqrt(x)
#else
#define LD_RSQRT(x) (1.0 / sqrt(x))
#endif

namespace liodom_dev {

constexpr int kJacobiSweeps = 8;          // fixed sweep count of the 3x3 eigenvalue iteration
constexpr int kLmMaxIterations = 4;       // src/laser_odometry.cc:214
constexpr double kHuberA = 0.2;           // src/laser_odometry.cc:201
constexpr double kPi = 3.14159265358979323846;  // M_PI

// finite test without libm / header dependencies (inf - inf and NaN - NaN are NaN)
LD_HD bool ld_isfinite(double x) { return (x - x) == 0.0; }

// ---------------------------------------------------------------------------------------
// Ring split (src/feature_extractor.cc:84-179)
// ---------------------------------------------------------------------------------------
// isValidPoint (:84-102).  dist is the XY range.
LD_HD bool valid_point(double x, double y, double z, double min_range, double max_range,
                       double* dist) {
  bool valid = true;
  if (!(ld_isfinite(x) && ld_isfinite(y) && ld_isfinite(z))) valid = false;
  *dist = sqrt(x * x + y * y);
  if (*dist > max_range || *dist < min_range) valid = false;
  return valid;
}

// Velodyne elevation binning (:127-151).  Returns -1 when the point is dropped.
LD_HD int velodyne_ring_from_angle(double angle, int scan_lines) {
  int scan_id;
  if (scan_lines == 64) {
    if (angle >= -8.83) scan_id = int((2 - angle) * 3.0 + 0.5);
    else scan_id = scan_lines / 2 + int((-8.83 - angle) * 2.0 + 0.5);
    if (angle > 2 || angle < -24.33 || scan_id > 63 || scan_id < 0) return -1;
  } else if (scan_lines == 32) {
    scan_id = int((angle + 92.0 / 3.0) * 3.0 / 4.0);
    if (scan_id > (scan_lines - 1) || scan_id < 0) return -1;
  } else if (scan_lines == 16) {
    scan_id = int((angle + 15) / 2 + 0.5);
    if (scan_id > (scan_lines - 1) || scan_id < 0) return -1;
  } else {
    return -1;
  }
  return scan_id;
}
// The same binning evaluated in FLOAT, for k_classify's fast decision only: the caller evaluates it at angle - m and angle + m and
// trusts the result when both agree (every branch, truncation and drop test is monotone in the angle; float rounding of these
// expressions is below 2e-5 degrees, the margin m is 2e-4).
LD_HD int velodyne_ring_from_angle_f(float angle, int scan_lines) {
  int scan_id;
  if (scan_lines == 64) {
    if (angle >= -8.83f) scan_id = (int)((2.0f - angle) * 3.0f + 0.5f);
    else scan_id = scan_lines / 2 + (int)((-8.83f - angle) * 2.0f + 0.5f);
    if (angle > 2.0f || angle < -24.33f || scan_id > 63 || scan_id < 0) return -1;
  } else if (scan_lines == 32) {
    scan_id = (int)((angle + 92.0f / 3.0f) * 0.75f);
    if (scan_id > (scan_lines - 1) || scan_id < 0) return -1;
  } else if (scan_lines == 16) {
    scan_id = (int)((angle + 15.0f) * 0.5f + 0.5f);
    if (scan_id > (scan_lines - 1) || scan_id < 0) return -1;
  } else {
    return -1;
  }
  return scan_id;
}
LD_HD int velodyne_ring(double z, double dist, int scan_lines) {
  const double angle = atan(z / dist) * 180 / kPi;
  return velodyne_ring_from_angle(angle, scan_lines);
}

// ---------------------------------------------------------------------------------------
// Curvature stencil (src/feature_extractor.cc:196-229).  All eleven operands are
// pcl::PointXYZI floats and `10 * x` is int * float, so the reference sums in FLOAT, left to
// right, and only the finished sum is widened to double (`double diff_x = ...`); the three squares
// and their sum (:229) are double.  (The reference build is x86-64 SSE: float expressions are
// evaluated in float, FLT_EVAL_METHOD = 0.)
// ---------------------------------------------------------------------------------------
LD_HD double stencil_sum(float m5, float m4, float m3, float m2, float m1, float c, float p1,
                         float p2, float p3, float p4, float p5) {
  const float s = m5 + m4 + m3 + m2 + m1 - 10 * c + p1 + p2 + p3 + p4 + p5;
  return (double)s;
}
LD_HD double stencil_axis(const float* p, int j) {
  return stencil_sum(p[j - 5], p[j - 4], p[j - 3], p[j - 2], p[j - 1], p[j], p[j + 1], p[j + 2],
                     p[j + 3], p[j + 4], p[j + 5]);
}
LD_HD double curvature(const float* px, const float* py, const float* pz, int j) {
  const double dx = stencil_axis(px, j), dy = stencil_axis(py, j), dz = stencil_axis(pz, j);
  return dx * dx + dy * dy + dz * dz;
}
// squared gap between consecutive ring points (:281-289, :297-305): float differences
// (`double diff_x = p[i].x - p[k].x` subtracts two floats), squares and sum in double
LD_HD double gap_sq3(float xi, float yi, float zi, float xk, float yk, float zk) {
  const double dx = (double)(xi - xk);
  const double dy = (double)(yi - yk);
  const double dz = (double)(zi - zk);
  return dx * dx + dy * dy + dz * dz;
}
LD_HD double gap_sq(const float* px, const float* py, const float* pz, int i, int k) {
  return gap_sq3(px[i], py[i], pz[i], px[k], py[k], pz[k]);
}

// ---------------------------------------------------------------------------------------
// Pose helpers.  T = 3x4 row-major [R | t] (Eigen::Isometry3d::matrix() top rows).
// ---------------------------------------------------------------------------------------
LD_HD void iso_identity(double* T) {
  for (int i = 0; i < 12; i++) T[i] = 0.0;
  T[0] = T[5] = T[10] = 1.0;
}
// (fully unrolled: with run-time indices the local 3 x 4 arrays of the callers live in scratch memory on the GPU)
LD_HD void iso_mul(const double* A, const double* B, double* C) {
  LD_UNROLL
  for (int r = 0; r < 3; r++) {
    LD_UNROLL
    for (int c = 0; c < 3; c++)
      C[r * 4 + c] = A[r * 4 + 0] * B[0 * 4 + c] + A[r * 4 + 1] * B[1 * 4 + c] + A[r * 4 + 2] * B[2 * 4 + c];
    C[r * 4 + 3] = A[r * 4 + 0] * B[3] + A[r * 4 + 1] * B[7] + A[r * 4 + 2] * B[11] + A[r * 4 + 3];
  }
}
LD_HD void iso_inverse(const double* A, double* C) {
  LD_UNROLL
  for (int r = 0; r < 3; r++) {
    LD_UNROLL
    for (int c = 0; c < 3; c++) C[r * 4 + c] = A[c * 4 + r];
  }
  LD_UNROLL
  for (int r = 0; r < 3; r++)
    C[r * 4 + 3] = -(C[r * 4 + 0] * A[3] + C[r * 4 + 1] * A[7] + C[r * 4 + 2] * A[11]);
}
// Eigen::Quaterniond(Matrix3d) (src/laser_odometry.cc:186); q = [x y z w].  The branch on the
// largest diagonal element is spelled out per case: run-time indices would push T and q to scratch
// memory on the GPU.
template <int I>
LD_HD void quat_from_rot_case(const double* T, double* q) {
  constexpr int J = (I + 1) % 3, K = (J + 1) % 3;
  double t = sqrt(T[I * 4 + I] - T[J * 4 + J] - T[K * 4 + K] + 1.0);
  q[I] = 0.5 * t;
  t = 0.5 / t;
  q[3] = (T[K * 4 + J] - T[J * 4 + K]) * t;
  q[J] = (T[J * 4 + I] + T[I * 4 + J]) * t;
  q[K] = (T[K * 4 + I] + T[I * 4 + K]) * t;
}
LD_HD void quat_from_rot(const double* T, double* q) {
  double t = T[0] + T[5] + T[10];
  if (t > 0.0) {
    t = sqrt(t + 1.0);
    q[3] = 0.5 * t;
    t = 0.5 / t;
    q[0] = (T[9] - T[6]) * t;
    q[1] = (T[2] - T[8]) * t;
    q[2] = (T[4] - T[1]) * t;
  } else {
    int i = 0;
    if (T[5] > T[0]) i = 1;
    if (T[10] > (i == 0 ? T[0] : T[5])) i = 2;
    if (i == 0) quat_from_rot_case<0>(T, q);
    else if (i == 1) quat_from_rot_case<1>(T, q);
    else quat_from_rot_case<2>(T, q);
  }
}
// Eigen::Quaterniond::toRotationMatrix + translation (src/laser_odometry.cc:225-227)
LD_HD void iso_from_qt(const double* q, const double* t, double* T) {
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w;
  const double txx = tx * x, txy = ty * x, txz = tz * x;
  const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
  T[0] = 1 - (tyy + tzz); T[1] = txy - twz;       T[2] = txz + twy;        T[3] = t[0];
  T[4] = txy + twz;       T[5] = 1 - (txx + tzz); T[6] = tyz - twx;        T[7] = t[1];
  T[8] = txz - twy;       T[9] = tyz + twx;       T[10] = 1 - (txx + tyy); T[11] = t[2];
}
// Eigen::Transform::rotation() (src/laser_odometry.cc:164,186,403,420).  Eigen 3.3.x (the
// README's platform: Ubuntu 20.04 ships 3.3.7) implements it for every Mode as
// computeRotationScaling(): JacobiSVD of linear(), R = U V^T (made proper) — the orthonormal polar
// factor.  Eigen >= 3.4 returns linear() itself for an Isometry.  mode 1 = polar factor, 0 = linear().
// The polar factor is computed by Newton's iteration X <- (X + X^-T) / 2 (quadratic; the poses on
// this path are orthonormal to ~1e-8 or better, so two steps reach rounding level); the oracle uses
// a Jacobi SVD — the factor is unique, both agree to ~1e-16.  det(linear) > 0 is assumed (always
// true for matrices built by toRotationMatrix and their products).
LD_HD void rotation_of(const double* T, int mode, double* R /*3x4, translation copied*/) {
  double x[9] = {T[0], T[1], T[2], T[4], T[5], T[6], T[8], T[9], T[10]};
  if (mode != 0) {
    for (int it = 0; it < 24; it++) {
      double c[9];
      c[0] = x[4] * x[8] - x[5] * x[7]; c[1] = x[5] * x[6] - x[3] * x[8]; c[2] = x[3] * x[7] - x[4] * x[6];
      c[3] = x[2] * x[7] - x[1] * x[8]; c[4] = x[0] * x[8] - x[2] * x[6]; c[5] = x[1] * x[6] - x[0] * x[7];
      c[6] = x[1] * x[5] - x[2] * x[4]; c[7] = x[2] * x[3] - x[0] * x[5]; c[8] = x[0] * x[4] - x[1] * x[3];
      const double det = x[0] * c[0] + x[1] * c[1] + x[2] * c[2];
      if (!(det > 0.0) || !ld_isfinite(det)) break;          // singular / improper / NaN: leave as is
      // far from orthonormal (never on a healthy trajectory): Frobenius scaling keeps the iteration fast
      double nx = 0.0, nc = 0.0;
      LD_UNROLL
      for (int i = 0; i < 9; i++) { nx += x[i] * x[i]; nc += c[i] * c[i]; }
      double g = 1.0;
      if (fabs(nx - 3.0) > 0.1) g = sqrt(sqrt(nc / (det * det)) / sqrt(nx));   // sqrt(|X^-1|_F / |X|_F)
      const double a = 0.5 * g, b = 0.5 / (g * det);
      double change = 0.0;
      LD_UNROLL
      for (int i = 0; i < 9; i++) {
        const double nv = a * x[i] + b * c[i];
        const double d = fabs(nv - x[i]);
        if (d > change) change = d;
        x[i] = nv;
      }
      if (change <= 4e-16) break;
    }
  }
  R[0] = x[0]; R[1] = x[1]; R[2] = x[2]; R[3] = T[3];
  R[4] = x[3]; R[5] = x[4]; R[6] = x[5]; R[7] = T[7];
  R[8] = x[6]; R[9] = x[7]; R[10] = x[8]; R[11] = T[11];
}
// Eigen::Quaterniond(T.rotation())
LD_HD void quat_from_pose(const double* T, int rotation_mode, double* q) {
  if (rotation_mode == 0) { quat_from_rot(T, q); return; }
  double R[12];
  rotation_of(T, rotation_mode, R);
  quat_from_rot(R, q);
}
// pcl::transformPointCloud, double matrix, float points (src/laser_odometry.cc:232,308)
LD_HD void transform_point(const double* T, float x, float y, float z, float* ox, float* oy,
                           float* oz) {
  const double dx = x, dy = y, dz = z;
  *ox = (float)(T[0] * dx + T[1] * dy + T[2] * dz + T[3]);
  *oy = (float)(T[4] * dx + T[5] * dy + T[6] * dz + T[7]);
  *oz = (float)(T[8] * dx + T[9] * dy + T[10] * dz + T[11]);
}
// ---- IMU roll / pitch override of the prediction (src/laser_odometry.cc:152-183) ----
// tf::Matrix3x3 (ROS tf LinearMath = Bullet btMatrix3x3, tfScalar = double): setRotation, getRPY
// (getEulerYPR solution 1), setRPY (setEulerYPR), getRotation.  3 x 3 row-major.
LD_HD void tf_matrix_from_quat(const double* q, double* m) {
  const double d = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  const double s = 2.0 / d;
  const double xs = q[0] * s, ys = q[1] * s, zs = q[2] * s;
  const double wx = q[3] * xs, wy = q[3] * ys, wz = q[3] * zs;
  const double xx = q[0] * xs, xy = q[0] * ys, xz = q[0] * zs;
  const double yy = q[1] * ys, yz = q[1] * zs, zz = q[2] * zs;
  m[0] = 1.0 - (yy + zz); m[1] = xy - wz;         m[2] = xz + wy;
  m[3] = xy + wz;         m[4] = 1.0 - (xx + zz); m[5] = yz - wx;
  m[6] = xz - wy;         m[7] = yz + wx;         m[8] = 1.0 - (xx + yy);
}
LD_HD void tf_get_rpy(const double* m, double* roll, double* pitch, double* yaw) {
  const double kPi = 3.14159265358979323846;
  if (fabs(m[6]) >= 1) {            // gimbal lock
    *yaw = 0;
    *roll = atan2(m[7], m[8]);
    *pitch = m[6] < 0 ? kPi / 2.0 : -kPi / 2.0;
  } else {
    *pitch = -asin(m[6]);
    const double cp = cos(*pitch);
    *roll = atan2(m[7] / cp, m[8] / cp);
    *yaw = atan2(m[3] / cp, m[0] / cp);
  }
}
LD_HD void tf_set_rpy(double roll, double pitch, double yaw, double* m) {
  const double ci = cos(roll), cj = cos(pitch), ch = cos(yaw);
  const double si = sin(roll), sj = sin(pitch), sh = sin(yaw);
  const double cc = ci * ch, cs = ci * sh, sc = si * ch, ss = si * sh;
  m[0] = cj * ch; m[1] = sj * sc - cs; m[2] = sj * cc + ss;
  m[3] = cj * sh; m[4] = sj * ss + cc; m[5] = sj * cs - sc;
  m[6] = -sj;     m[7] = cj * si;      m[8] = cj * ci;
}
LD_HD void tf_quat_from_matrix(const double* m, double* q) {
  const double trace = m[0] + m[4] + m[8];
  if (trace > 0.0) {
    double s = sqrt(trace + 1.0);
    q[3] = s * 0.5;
    s = 0.5 / s;
    q[0] = (m[7] - m[5]) * s;
    q[1] = (m[2] - m[6]) * s;
    q[2] = (m[3] - m[1]) * s;
  } else {
    const int i = m[0] < m[4] ? (m[4] < m[8] ? 2 : 1) : (m[0] < m[8] ? 2 : 0);
    const int j = (i + 1) % 3, k = (i + 2) % 3;
    double s = sqrt(m[i * 3 + i] - m[j * 3 + j] - m[k * 3 + k] + 1.0);
    q[i] = s * 0.5;
    s = 0.5 / s;
    q[3] = (m[k * 3 + j] - m[j * 3 + k]) * s;
    q[j] = (m[j * 3 + i] + m[i * 3 + j]) * s;
    q[k] = (m[k * 3 + i] + m[i * 3 + k]) * s;
  }
}
// odom (3 x 4, world <- laser) with the roll and pitch of its base_link orientation replaced by the IMU's
LD_HD void imu_override(const double* odom, const double* imu_q, const double* laser_to_base, int rotation_mode, double* out) {
  double m[9], imu_roll, imu_pitch, imu_yaw, bl_roll, bl_pitch, bl_yaw;
  tf_matrix_from_quat(imu_q, m);
  tf_get_rpy(m, &imu_roll, &imu_pitch, &imu_yaw);                 // :155-161
  double odom_bl[12], q_bl[4], q_new[4], l2b_inv[12];
  iso_mul(odom, laser_to_base, odom_bl);                          // :164
  quat_from_pose(odom_bl, rotation_mode, q_bl);                   // :165 Quaterniond(odom_bl.rotation())
  tf_matrix_from_quat(q_bl, m);
  tf_get_rpy(m, &bl_roll, &bl_pitch, &bl_yaw);                    // :166-169
  tf_set_rpy(imu_roll, imu_pitch, bl_yaw, m);                     // :174
  tf_quat_from_matrix(m, q_new);                                  // :177
  const double t_bl[3] = {odom_bl[3], odom_bl[7], odom_bl[11]};
  iso_from_qt(q_new, t_bl, odom_bl);                              // :178-179
  iso_inverse(laser_to_base, l2b_inv);
  iso_mul(odom_bl, l2b_inv, out);                                 // :182
}
// LaserOdometer::publishOdom (src/laser_odometry.cc:395-436): pose in the base_link frame and the
// finite-difference twist.  out: orientation x y z w, position, twist.linear, twist.angular (13).
LD_HD void odom_message(const double* prev_odom, const double* odom, const double* laser_to_base,
                        double delta_time, int rotation_mode, double* out) {
  double bl[12], pbl[12], pinv[12], d[12], qd[4], m[9], roll, pitch, yaw;
  iso_mul(odom, laser_to_base, bl);                 // :403
  quat_from_pose(bl, rotation_mode, out);           // :403 q_current(odom_base_link.rotation())
  out[4] = bl[3]; out[5] = bl[7]; out[6] = bl[11];  // :405
  iso_mul(prev_odom, laser_to_base, pbl);
  iso_inverse(pbl, pinv);
  iso_mul(pinv, bl, d);                             // :416
  out[7] = d[3] / delta_time; out[8] = d[7] / delta_time; out[9] = d[11] / delta_time;   // :417-420
  quat_from_pose(d, rotation_mode, qd);             // :420 q_delta(delta_odom.rotation())
  tf_matrix_from_quat(qd, m);
  tf_get_rpy(m, &roll, &pitch, &yaw);               // :423-426
  out[10] = roll / delta_time; out[11] = pitch / delta_time; out[12] = yaw / delta_time;
}
// ceres::EigenQuaternionParameterization::Plus, x = [x y z w]
LD_HD void quat_plus(const double* x, const double* delta, double* out) {
  LD_FP_CONTRACT_FAST
  const double u2 = delta[0] * delta[0] + delta[1] * delta[1] + delta[2] * delta[2];
  if (u2 > 0.0) {
    // sin(nd)/nd and cos(nd).  LM steps are small rotations: below 0.5 rad the Taylor series in
    // nd^2 (9 terms, truncation error < 1e-19) replaces the two ~170-instruction library calls on
    // the single lane that runs the controller — and needs no square root (nd only enters as nd^2);
    // larger steps take the library path.
    double s, aw;
    if (u2 < 0.25) {
      const double u = u2;
      s = 1.0 + u * (-1.0 / 6.0 + u * (1.0 / 120.0 + u * (-1.0 / 5040.0 + u * (1.0 / 362880.0 + u * (-1.0 / 39916800.0 +
          u * (1.0 / 6227020800.0 + u * (-1.0 / 1307674368000.0 + u * (1.0 / 355687428096000.0))))))));
      aw = 1.0 + u * (-0.5 + u * (1.0 / 24.0 + u * (-1.0 / 720.0 + u * (1.0 / 40320.0 + u * (-1.0 / 3628800.0 +
           u * (1.0 / 479001600.0 + u * (-1.0 / 87178291200.0 + u * (1.0 / 20922789888000.0))))))));
    } else {
      const double nd = sqrt(u2);
      s = sin(nd) / nd;
      aw = cos(nd);
    }
    const double ax = s * delta[0], ay = s * delta[1], az = s * delta[2];
    const double bw = x[3], bx = x[0], by = x[1], bz = x[2];
    out[3] = aw * bw - ax * bx - ay * by - az * bz;
    out[0] = aw * bx + ax * bw + ay * bz - az * by;
    out[1] = aw * by + ay * bw + az * bx - ax * bz;
    out[2] = aw * bz + az * bw + ax * by - ay * bx;
  } else {
    out[0] = x[0]; out[1] = x[1]; out[2] = x[2]; out[3] = x[3];
  }
}

// ---------------------------------------------------------------------------------------
// kNN distance, FLANN L2_Simple<float> order
// ---------------------------------------------------------------------------------------
LD_HD float sqdist_f(float qx, float qy, float qz, float mx, float my, float mz) {
  const float dx = qx - mx, dy = qy - my, dz = qz - mz;
  float r = dx * dx;
  r = r + dy * dy;
  r = r + dz * dz;
  return r;
}

// ---------------------------------------------------------------------------------------
// Line gate (src/laser_odometry.cc:325-344): centroid and scatter matrix of the five nearest
// neighbours in FP64, eigenvalues by kJacobiSweeps sweeps of cyclic Jacobi (only + - * / sqrt),
// accept iff lambda_max > 3 * lambda_mid.
// ---------------------------------------------------------------------------------------
LD_HD void jacobi_rot(double& app, double& aqq, double& apq, double& arp, double& arq) {
  // converged pair: rotating further cannot change the eigenvalues at double precision (the
  // same test, bit for bit, in the oracle and in the GPU path)
  if (fabs(apq) <= 1e-20 * (fabs(app) + fabs(aqq))) return;
  const double theta = (aqq - app) / (2.0 * apq);
  const double at = fabs(theta);
  double t = 1.0 / (at + sqrt(theta * theta + 1.0));
  if (theta < 0.0) t = -t;
  const double c = 1.0 / sqrt(t * t + 1.0);
  const double s = t * c;
  app = app - t * apq;
  aqq = aqq + t * apq;
  apq = 0.0;
  const double nrp = c * arp - s * arq;
  const double nrq = s * arp + c * arq;
  arp = nrp; arq = nrq;
}
LD_HD void eig3_sym(const double* a, double* ev) {
  double a00 = a[0], a01 = a[1], a02 = a[2], a11 = a[3], a12 = a[4], a22 = a[5];
  for (int sweep = 0; sweep < kJacobiSweeps; sweep++) {
    jacobi_rot(a00, a11, a01, a02, a12);
    jacobi_rot(a00, a22, a02, a01, a12);
    jacobi_rot(a11, a22, a12, a01, a02);
  }
  double e0 = a00, e1 = a11, e2 = a22, s;
  if (e0 > e1) { s = e0; e0 = e1; e1 = s; }
  if (e1 > e2) { s = e1; e1 = e2; e2 = s; }
  if (e0 > e1) { s = e0; e0 = e1; e1 = s; }
  ev[0] = e0; ev[1] = e1; ev[2] = e2;
}
// nn = 5 points (x,y,z) as floats, in ascending-distance order
LD_HD bool line_gate(const float* nx, const float* ny, const float* nz) {
  double cx = 0, cy = 0, cz = 0;
  for (int j = 0; j < 5; j++) { cx = cx + (double)nx[j]; cy = cy + (double)ny[j]; cz = cz + (double)nz[j]; }
  cx = cx / 5.0; cy = cy / 5.0; cz = cz / 5.0;
  double cov[6] = {0, 0, 0, 0, 0, 0};
  for (int j = 0; j < 5; j++) {
    const double zx = (double)nx[j] - cx, zy = (double)ny[j] - cy, zz = (double)nz[j] - cz;
    cov[0] = cov[0] + zx * zx; cov[1] = cov[1] + zx * zy; cov[2] = cov[2] + zx * zz;
    cov[3] = cov[3] + zy * zy; cov[4] = cov[4] + zy * zz; cov[5] = cov[5] + zz * zz;
  }
  // The gate needs a decision, not eigenvalues.  The closed-form (trigonometric) eigenvalues of a symmetric
  // 3 x 3 matrix decide unless lambda2 - 3*lambda1 is too close to zero for their accuracy; the (rare) rest goes
  // through the iterative solver (the oracle's algorithm).  Two stages, because the gate runs on one lane per
  // query while the rest of its wave waits, so its instruction count is paid by the whole wave:
  //   (1) float trigonometry (atan2f / cosf: ~110 instructions): the angle is formed from sqrt((1-r)(1+r)) and r,
  //       which is well conditioned near |r| = 1 (the line-like case), so the eigenvalues are good to ~1e-6
  //       relative; decides unless |diff| <= 1e-4 * lambda_max;
  //   (2) double trigonometry (acos / cos: ~330 instructions): decides unless |diff| <= 1e-9 * lambda_max.
  const double p1 = cov[1] * cov[1] + cov[2] * cov[2] + cov[4] * cov[4];
  const double q = (cov[0] + cov[3] + cov[5]) / 3.0;
  const double d0 = cov[0] - q, d1 = cov[3] - q, d2 = cov[5] - q;
  const double p2 = d0 * d0 + d1 * d1 + d2 * d2 + 2.0 * p1;
  const double pp = sqrt(p2 / 6.0);
  if (pp > 0.0 && ld_isfinite(pp)) {
    const double ip = 1.0 / pp;
    const double b00 = d0 * ip, b11 = d1 * ip, b22 = d2 * ip, b01 = cov[1] * ip, b02 = cov[2] * ip, b12 = cov[4] * ip;
    double r = 0.5 * (b00 * (b11 * b22 - b12 * b12) - b01 * (b01 * b22 - b12 * b02) + b02 * (b01 * b12 - b11 * b02));
    r = r < -1.0 ? -1.0 : (r > 1.0 ? 1.0 : r);
    {
      const float s2 = (float)((1.0 - r) * (1.0 + r));
      const float phi = atan2f(sqrtf(s2 > 0.f ? s2 : 0.f), (float)r) * (1.0f / 3.0f);
      const double c_max = (double)cosf(phi), c_min = (double)cosf(phi + 2.0943951f);
      const double e_max = q + 2.0 * pp * c_max;
      const double e_min = q + 2.0 * pp * c_min;
      const double e_mid = 3.0 * q - e_max - e_min;
      const double diff = e_max - 3.0 * e_mid;
      if (fabs(diff) > 1e-4 * e_max) return diff > 0.0;
    }
    const double phi = acos(r) / 3.0;
    const double e_max = q + 2.0 * pp * cos(phi);
    const double e_min = q + 2.0 * pp * cos(phi + 2.0943951023931954923);   // + 2 pi / 3
    const double e_mid = 3.0 * q - e_max - e_min;
    const double diff = e_max - 3.0 * e_mid;
    if (fabs(diff) > 1e-9 * e_max) return diff > 0.0;
  }
  double ev[3];
  eig3_sym(cov, ev);
  return ev[2] > 3 * ev[1];
}

// ---------------------------------------------------------------------------------------
// Point-to-line residual block (include/liodom/factors.hpp:71-105) with the analytic tangent
// Jacobian of SURVEY.md A.4 and the Huber(0.2) correction, accumulated into the normal
// equations.  Accumulator layout: v[0] = cost (0.5*sum rho), v[1..6] = g = J^T r,
// v[7..27] = upper triangle of J^T J (row-major: 00 01 .. 05 11 12 .. 55), v[28] = number of
// non-finite blocks.
// ---------------------------------------------------------------------------------------
constexpr int kAccN = 29;

LD_HD int h_idx(int i, int j) {  // upper-triangle index, i <= j
  return i * 6 - (i * (i - 1)) / 2 + (j - i);
}
LD_HD double h_at(const double* H, int i, int j) { return i <= j ? H[h_idx(i, j)] : H[h_idx(j, i)]; }

// One block: the loss-corrected quantities the normal equations are built from.
//   J[18]  3 x 6 tangent Jacobian (row-major: residual r, column c at J[6 r + c]), NOT yet scaled by sqrt(rho')
//   rs[3]  rho' * residual,  rho0 = rho(|r|^2),  rho1 = rho'
// Returns false if anything is non-finite (ceres IsEvaluationValid).
LD_HD bool residual_block(const double* Rm /*3x4*/, const double* p, const double* a, const double* b,
                          double min_d, double max_d, double* J, double* rs, double* rho0_out, double* rho1_out) {
  LD_FP_CONTRACT_FAST
  const double tx = Rm[3], ty = Rm[7], tz = Rm[11];
  const double Rp0 = Rm[0] * p[0] + Rm[1] * p[1] + Rm[2] * p[2];
  const double Rp1 = Rm[4] * p[0] + Rm[5] * p[1] + Rm[6] * p[2];
  const double Rp2 = Rm[8] * p[0] + Rm[9] * p[1] + Rm[10] * p[2];
  const double lp0 = Rp0 + tx, lp1 = Rp1 + ty, lp2 = Rp2 + tz;
  const double u0 = lp0 - a[0], u1 = lp1 - a[1], u2 = lp2 - a[2];
  const double w0 = lp0 - b[0], w1 = lp1 - b[1], w2 = lp2 - b[2];
  const double nu0 = u1 * w2 - u2 * w1, nu1 = u2 * w0 - u0 * w2, nu2 = u0 * w1 - u1 * w0;
  const double de0 = a[0] - b[0], de1 = a[1] - b[1], de2 = a[2] - b[2];
  const double invL = LD_RSQRT(de0 * de0 + de1 * de1 + de2 * de2);      // 1 / ||de||  (factors.hpp:100)
  const double cx = p[0] - tx, cy = p[1] - ty;
  const double rho_sq = cx * cx + cy * cy;
  const double inv_rho = LD_RSQRT(rho_sq);
  const double rho = rho_sq * inv_rho;                                  // (0 * inf = NaN at rho = 0, like 0 / 0 below)
  const double inv_range = 1.0 / (max_d - min_d);                       // uniform: hoisted out of the loop
  const double w = 1.01 - (rho - min_d) * inv_range;
  const double wl = w * invL;
  const double n0 = nu0 * invL, n1 = nu1 * invL, n2 = nu2 * invL;
  const double r0 = w * n0, r1 = w * n1, r2 = w * n2;              // factors.hpp:99-101
  const double s = r0 * r0 + r1 * r1 + r2 * r2;
  // Jq = (2w/L) [de]x [Rp]x ; [de]x[Rp]x = Rp de^T - (de.Rp) I
  const double dot = de0 * Rp0 + de1 * Rp1 + de2 * Rp2;
  const double k2 = 2.0 * wl;
  J[0]  = k2 * (Rp0 * de0 - dot); J[1]  = k2 * (Rp0 * de1);       J[2]  = k2 * (Rp0 * de2);
  J[6]  = k2 * (Rp1 * de0);       J[7]  = k2 * (Rp1 * de1 - dot); J[8]  = k2 * (Rp1 * de2);
  J[12] = k2 * (Rp2 * de0);       J[13] = k2 * (Rp2 * de1);       J[14] = k2 * (Rp2 * de2 - dot);
  // Jt = -(w/L)[de]x + (nu/L) (dw/dt)^T,  dw/dt = (cx, cy, 0) / (rho * range)
  const double inv_rr = inv_rho * inv_range;
  const double dwx = cx * inv_rr, dwy = cy * inv_rr;
  J[3]  = n0 * dwx;              J[4]  = wl * de2 + n0 * dwy;   J[5]  = -wl * de1;
  J[9]  = -wl * de2 + n1 * dwx;  J[10] = n1 * dwy;              J[11] = wl * de0;
  J[15] = wl * de1 + n2 * dwx;   J[16] = -wl * de0 + n2 * dwy;  J[17] = 0.0;
  // Huber (ceres::HuberLoss): rho'' <= 0 -> residual and Jacobian scaled by sqrt(rho')
  double rho0, rho1;
  const double bsq = kHuberA * kHuberA;
  if (s > bsq) {
    const double inv_rr2 = LD_RSQRT(s);
    rho0 = 2.0 * kHuberA * (s * inv_rr2) - bsq;
    rho1 = kHuberA * inv_rr2;
    if (rho1 < DBL_MIN) rho1 = DBL_MIN;
  } else {
    rho0 = s; rho1 = 1.0;
  }
  // any non-finite residual or Jacobian entry invalidates the block (ceres IsEvaluationValid):
  // x * 0 is NaN exactly for x = +-inf / NaN, so one FMA chain + one compare test all 19 values
  double z = s * 0.0;
  LD_UNROLL
  for (int i = 0; i < 18; i++) z += J[i] * 0.0;
  *rho0_out = rho0; *rho1_out = rho1;
  rs[0] = rho1 * r0; rs[1] = rho1 * r1; rs[2] = rho1 * r2;
  return z == 0.0;
}
// Entry e of the 29-entry accumulator contributed by one valid block (see residual_accumulate for the layout).
LD_HD double residual_entry(const double* J, const double* rs, double rho0, double rho1, int e) {
  LD_FP_CONTRACT_FAST
  if (e == 0) return 0.5 * rho0;
  if (e <= 6) { const int i = e - 1; return J[i] * rs[0] + J[6 + i] * rs[1] + J[12 + i] * rs[2]; }
  if (e >= 28) return 0.0;
  int k = e - 7, i = 0;                       // upper-triangle index -> (i, j)
  while (k >= 6 - i) { k -= 6 - i; i++; }
  const int j = i + k;
  const double a0 = rho1 * J[i], a1 = rho1 * J[6 + i], a2 = rho1 * J[12 + i];
  return a0 * J[j] + a1 * J[6 + j] + a2 * J[12 + j];
}

LD_HD void residual_accumulate(const double* Rm /*3x4*/, const double* p, const double* a,
                               const double* b, double min_d, double max_d, double* acc) {
  LD_FP_CONTRACT_FAST
  double J[18], rs[3], rho0, rho1;
  const bool ok = residual_block(Rm, p, a, b, min_d, max_d, J, rs, &rho0, &rho1);
  // (both counters updated unconditionally, by selects: as `if (!ok) { acc[28] += 1; return; } acc[0] += ...` the compiler merged
  //  the two updates into ONE add at a run-time index (ok ? 0 : 28) — which put the accumulator array into scratch memory and a
  //  scratch load + store into every block evaluation of k_lm_solve)
  acc[28] += ok ? 0.0 : 1.0;
  acc[0] += ok ? 0.5 * rho0 : 0.0;
  if (!ok) return;
  LD_UNROLL
  for (int i = 0; i < 6; i++) {
    acc[1 + i] += J[i] * rs[0] + J[6 + i] * rs[1] + J[12 + i] * rs[2];
    const double a0 = rho1 * J[i], a1 = rho1 * J[6 + i], a2 = rho1 * J[12 + i];
    LD_UNROLL
    for (int j = i; j < 6; j++) acc[7 + h_idx(i, j)] += a0 * J[j] + a1 * J[6 + j] + a2 * J[12 + j];
  }
}

// ---------------------------------------------------------------------------------------
// Trust-region Levenberg-Marquardt controller mirroring ceres::Solve as configured at
// src/laser_odometry.cc:212-218 (Ceres <= 2.1 defaults): Jacobi scaling from the
// first Jacobian, radius 1e4, diagonal clamp [1e-6, 1e32], step quality > 1e-3, parameter
// tolerance 1e-8, function tolerance 1e-6, gradient tolerance 1e-10, max 4 iterations where
// every trial counts.  DENSE_QR on the stacked system is replaced by a Cholesky solve of the
// 6x6 normal equations (same minimiser).
//
// Usage: lm_begin(st, x0, acc0) -> if LM_NEED_EVAL evaluate acc at st.cand_* and call
// lm_update(st, acc) until LM_DONE.  The solution is st.q / st.t.
// ---------------------------------------------------------------------------------------
enum { LM_DONE = 0, LM_NEED_EVAL = 1 };
enum {
  LM_TERM_MAX_ITER = 0, LM_TERM_PARAM_TOL = 1, LM_TERM_FUNC_TOL = 2, LM_TERM_GRAD_TOL = 3,
  LM_TERM_NO_RESIDUALS = 4, LM_TERM_EVAL_FAILURE = 5, LM_TERM_RADIUS = 6, LM_TERM_INVALID_STEPS = 7
};

struct LmState {
  double q[4], t[3];            // current (accepted) point
  double cand_q[4], cand_t[3];  // candidate to evaluate
  double cost;                  // cost at the current point
  double g[6], H[21];           // J^T r and J^T J (unscaled, loss-corrected) at the current point
  double scale[6];              // Jacobi scaling, fixed per solve
  double diag[6];
  double radius, decrease_factor;
  double x_norm;
  double model_cost_change;
  double initial_cost;
  int reuse_diagonal;
  int iter, accepted, invalid_run, termination;
  int apply_on_ftol;
};


// Cholesky solve of the 6x6 SPD system A y = b, IN PLACE on the packed lower triangle (row-major: A[lt(i, j)], j <= i; the
// factor overwrites it).  Returns false if not positive definite.  (The controller runs on one lane of a kernel whose other
// lanes hold 256 registers of evaluation state: three 6 x 6 work matrices — scaled H, H + D, factor — were spilled around
// every step; this keeps 21 doubles.)
LD_HD int lt_idx(int i, int j) { return i * (i + 1) / 2 + j; }      // j <= i
LD_HD bool chol_solve6_packed(double* A /*21, overwritten by L*/, const double* b, double* y) {
  LD_FP_CONTRACT_FAST
  double inv[6];             // reciprocals of the diagonal: one division per column
  LD_UNROLL
  for (int j = 0; j < 6; j++) {
    double d = A[lt_idx(j, j)];
    LD_UNROLL
    for (int k = 0; k < j; k++) d -= A[lt_idx(j, k)] * A[lt_idx(j, k)];
    if (!(d > 0.0) || !ld_isfinite(d)) return false;
#if defined(__HIP_DEVICE_COMPILE__)
    inv[j] = rsqrt(d);             // only the reciprocal of the pivot is ever used
#else
    inv[j] = 1.0 / sqrt(d);
#endif
    LD_UNROLL
    for (int i = j + 1; i < 6; i++) {
      double s = A[lt_idx(i, j)];
      LD_UNROLL
      for (int k = 0; k < j; k++) s -= A[lt_idx(i, k)] * A[lt_idx(j, k)];
      A[lt_idx(i, j)] = s * inv[j];
    }
  }
  double z[6];
  LD_UNROLL
  for (int i = 0; i < 6; i++) {
    double s = b[i];
    LD_UNROLL
    for (int k = 0; k < i; k++) s -= A[lt_idx(i, k)] * z[k];
    z[i] = s * inv[i];
  }
  LD_UNROLL
  for (int i = 5; i >= 0; i--) {
    double s = z[i];
    LD_UNROLL
    for (int k = i + 1; k < 6; k++) s -= A[lt_idx(k, i)] * y[k];
    y[i] = s * inv[i];
    if (!ld_isfinite(y[i])) return false;
  }
  return true;
}

LD_HD double norm7(const double* q, const double* t) {
  LD_FP_CONTRACT_FAST
  return sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3] + t[0] * t[0] + t[1] * t[1] + t[2] * t[2]);
}

// Computes the next valid trust-region step and the candidate point.  Invalid steps shrink the
// radius and consume an iteration each (TrustRegionMinimizer::HandleInvalidStep).
LD_HD int lm_propose(LmState& st) {
  // The controller is not on the bit-exact path (pose tolerance 1e-4, and the oracle solves the step by QR anyway):
  // FMA contraction is allowed here, and only the lower triangle of the symmetric matrices is formed.  It runs on a
  // single lane 3-4 times per solve, so its instruction count is on the scan's critical path.
  LD_FP_CONTRACT_FAST
  while (true) {
    if (st.iter >= kLmMaxIterations) { st.termination = LM_TERM_MAX_ITER; return LM_DONE; }
    if (st.radius < 1e-32) { st.termination = LM_TERM_RADIUS; return LM_DONE; }
    st.iter++;
    double sc[6], gs[6];
    LD_UNROLL
    for (int i = 0; i < 6; i++) { sc[i] = st.scale[i]; gs[i] = sc[i] * st.g[i]; }
    // scaled normal matrix, packed lower triangle; the same products are formed again for the model cost change below (from the
    // state, which is in LDS) instead of being kept alive — in registers that the factorisation needs — across the solve
    double A[21];
    LD_UNROLL
    for (int i = 0; i < 6; i++) {
      LD_UNROLL
      for (int j = 0; j <= i; j++) A[lt_idx(i, j)] = sc[i] * sc[j] * st.H[h_idx(j, i)];
    }
    if (!st.reuse_diagonal) {
      LD_UNROLL
      for (int j = 0; j < 6; j++) {
        double d = A[lt_idx(j, j)];
        if (d < 1e-6) d = 1e-6;
        if (d > 1e32) d = 1e32;
        st.diag[j] = d;
      }
    }
    const double inv_radius = 1.0 / st.radius;
    LD_UNROLL
    for (int j = 0; j < 6; j++) A[lt_idx(j, j)] += st.diag[j] * inv_radius;
    double y[6];
    const bool ok = chol_solve6_packed(A, gs, y);
    st.reuse_diagonal = 1;
    double mcc = 0.0;
    double step[6];
    if (ok) {
      double sg = 0.0, shs = 0.0;
      LD_UNROLL
      for (int i = 0; i < 6; i++) step[i] = -y[i];
      LD_UNROLL
      for (int i = 0; i < 6; i++) {
        sg += step[i] * gs[i];
        // step^T Hs step from the lower triangle: diagonal once, off-diagonal entries twice
        double row = 0.0;
        LD_UNROLL
        for (int j = 0; j < i; j++) row += (sc[i] * sc[j] * st.H[h_idx(j, i)]) * step[j];
        shs += step[i] * (2.0 * row + (sc[i] * sc[i] * st.H[h_idx(i, i)]) * step[i]);
      }
      mcc = -sg - 0.5 * shs;
    }
    if (!ok || !(mcc > 0.0)) {
      // TrustRegionMinimizer::HandleInvalidStep (max_num_consecutive_invalid_steps = 5), then
      // LevenbergMarquardtStrategy::StepIsInvalid(): radius *= 0.5, reuse_diagonal = true;
      // decrease_factor belongs to StepRejected / StepAccepted only
      if (++st.invalid_run >= 5) { st.termination = LM_TERM_INVALID_STEPS; return LM_DONE; }
      st.radius = st.radius * 0.5;
      continue;
    }
    st.invalid_run = 0;
    st.model_cost_change = mcc;
    double delta[6];
    LD_UNROLL
    for (int j = 0; j < 6; j++) delta[j] = step[j] * sc[j];
    quat_plus(st.q, delta, st.cand_q);
    LD_UNROLL
    for (int k = 0; k < 3; k++) st.cand_t[k] = st.t[k] + delta[3 + k];
    return LM_NEED_EVAL;
  }
}

// acc = accumulator evaluated at (q0, t0); n_blocks = number of residual blocks.
// pre_scale (optional): the six Jacobi scales 1 / (1 + sqrt(H_jj)), computed by six lanes side by side on the device
LD_HD int lm_begin(LmState& st, const double* q0, const double* t0, const double* acc,
                   int n_blocks, int apply_on_ftol, const double* pre_scale = nullptr) {
  LD_UNROLL
  for (int k = 0; k < 4; k++) st.q[k] = st.cand_q[k] = q0[k];
  LD_UNROLL
  for (int k = 0; k < 3; k++) st.t[k] = st.cand_t[k] = t0[k];
  st.iter = 0; st.accepted = 0; st.invalid_run = 0; st.termination = LM_TERM_MAX_ITER;
  st.apply_on_ftol = apply_on_ftol;
  st.cost = 0.0; st.initial_cost = 0.0; st.model_cost_change = 0.0;
  st.radius = 1e4; st.decrease_factor = 2.0; st.reuse_diagonal = 0;
  LD_UNROLL
  for (int j = 0; j < 6; j++) { st.scale[j] = 1.0; st.diag[j] = 0.0; st.g[j] = 0.0; }
  LD_UNROLL
  for (int j = 0; j < 21; j++) st.H[j] = 0.0;
  st.x_norm = norm7(st.q, st.t);
  if (n_blocks == 0) { st.termination = LM_TERM_NO_RESIDUALS; return LM_DONE; }
  if (acc[28] != 0.0) { st.termination = LM_TERM_EVAL_FAILURE; return LM_DONE; }
  st.cost = acc[0]; st.initial_cost = acc[0];
  double gmax = 0.0;
  LD_UNROLL
  for (int j = 0; j < 6; j++) { st.g[j] = acc[1 + j]; const double ag = fabs(st.g[j]); if (ag > gmax) gmax = ag; }
  LD_UNROLL
  for (int j = 0; j < 21; j++) st.H[j] = acc[7 + j];
  LD_UNROLL
  for (int j = 0; j < 6; j++) st.scale[j] = pre_scale ? pre_scale[j] : 1.0 / (1.0 + sqrt(st.H[h_idx(j, j)]));
  if (gmax <= 1e-10) { st.termination = LM_TERM_GRAD_TOL; return LM_DONE; }
  return lm_propose(st);
}

// acc = accumulator evaluated at the candidate (cost + normal equations).
LD_HD int lm_update(LmState& st, const double* acc) {
  LD_FP_CONTRACT_FAST
  const double cand_cost = (acc[28] != 0.0) ? DBL_MAX : acc[0];
  double dq[4], dt[3];
  LD_UNROLL
  for (int k = 0; k < 4; k++) dq[k] = st.q[k] - st.cand_q[k];
  LD_UNROLL
  for (int k = 0; k < 3; k++) dt[k] = st.t[k] - st.cand_t[k];
  // parameter tolerance: Ceres tests step_norm <= parameter_tolerance * (x_norm + parameter_tolerance) with step_norm =
  // sqrt(step_sq) (the oracle has exactly that expression, oracle/liodom_oracle.cc lm_solve).  The squares are compared first — no
  // FP64 square root on the controller's lane, which is on the scan's critical path — and decide alone wherever the two forms
  // cannot differ: sqrt is monotone and correctly rounded, so step_sq <= ptol^2 (1 - 1e-12) implies sqrt(step_sq) < ptol and
  // step_sq > ptol^2 (1 + 1e-12) implies sqrt(step_sq) > ptol.  Inside that band (practically never) Ceres' own expression is
  // evaluated, so the decision is Ceres' in every case (tests/test_oracle_odometry.py::test_parameter_tolerance_decision_is_the_ceres_form,
  // tests/test_hostcheck.py::test_lm_update_on_the_parameter_tolerance_boundary).
  const double step_sq = dq[0] * dq[0] + dq[1] * dq[1] + dq[2] * dq[2] + dq[3] * dq[3] + dt[0] * dt[0] + dt[1] * dt[1] + dt[2] * dt[2];
  const double ptol = 1e-8 * (st.x_norm + 1e-8);
  const double ptol_sq = ptol * ptol;
  bool ptol_hit = step_sq <= ptol_sq * (1.0 - 1e-12);
  if (!ptol_hit && step_sq <= ptol_sq * (1.0 + 1e-12)) ptol_hit = sqrt(step_sq) <= ptol;
  if (ptol_hit) { st.termination = LM_TERM_PARAM_TOL; return LM_DONE; }
  const double cost_change = st.cost - cand_cost;
  if (fabs(cost_change) <= 1e-6 * st.cost) {
    st.termination = LM_TERM_FUNC_TOL;
    if (st.apply_on_ftol && cost_change > 0.0) {
      LD_UNROLL
      for (int k = 0; k < 4; k++) st.q[k] = st.cand_q[k];
      LD_UNROLL
      for (int k = 0; k < 3; k++) st.t[k] = st.cand_t[k];
      st.cost = cand_cost;
    }
    return LM_DONE;
  }
  const double rel = cost_change / st.model_cost_change;
  if (rel > 1e-3) {
    LD_UNROLL
    for (int k = 0; k < 4; k++) st.q[k] = st.cand_q[k];
    LD_UNROLL
    for (int k = 0; k < 3; k++) st.t[k] = st.cand_t[k];
    st.x_norm = norm7(st.q, st.t);
    st.cost = cand_cost;
    double gmax = 0.0;
    LD_UNROLL
    for (int j = 0; j < 6; j++) { st.g[j] = acc[1 + j]; const double ag = fabs(st.g[j]); if (ag > gmax) gmax = ag; }
    LD_UNROLL
    for (int j = 0; j < 21; j++) st.H[j] = acc[7 + j];
    st.accepted++;
    const double c = 2.0 * rel - 1.0;
    double f = 1.0 - c * c * c;
    if (f < 1.0 / 3.0) f = 1.0 / 3.0;
    st.radius = st.radius / f;
    if (st.radius > 1e16) st.radius = 1e16;
    st.decrease_factor = 2.0;
    st.reuse_diagonal = 0;
    if (gmax <= 1e-10) { st.termination = LM_TERM_GRAD_TOL; return LM_DONE; }
  } else {
    st.radius = st.radius / st.decrease_factor;
    st.decrease_factor *= 2.0;
    st.reuse_diagonal = 1;
  }
  return lm_propose(st);
}

}  // namespace liodom_dev
