// oracle/liodom_oracle.cc — CPU restatement of LiODOM's per-scan hot path.
//
// TEST INFRASTRUCTURE ONLY.  Nothing under liodom_amd/ (the product) may include, link, import or
// execute this file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it,
// and only as the checker / reported CPU baseline.
//
// PARITY UNPINNED.  The reference (/root/reference) has no tests, fixtures or golden vectors
//  and cannot be built in this image: every translation unit includes ROS, PCL,
// Eigen and Ceres headers that are absent, and writing stand-in headers is not allowed.  The
// third-party arithmetic (PCL 1.10 transformPointCloud / KdTreeFLANN / VoxelGrid, Eigen 3.3
// quaternion conversions, Ceres 1.14 trust-region LM — versions implied by README.md:38-41,
// not pinned by the reference) is restated from its published behaviour.
// What pins this file instead: an independent pure-Python transcription of the selection logic
// (tests/pyref.py), NumPy brute-force kNN / eigvalsh, finite-difference Jacobians and a SciPy
// minimiser on the same Huber cost (tests/test_oracle_*.py).
//
// Every function cites the reference file:line it follows (paths relative to /root/reference).
// Plain C++17, no dependencies.  Build with -ffp-contract=off (the reference build is x86-64
// -O3 without -march, i.e. no FMA; CMakeLists.txt:13).
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <deque>
#include <limits>
#include <map>
#include <queue>
#include <tuple>
#include <vector>

extern "C" {

// Mirrors the numeric fields of liodom::Params (include/liodom/params.h:33-49,
// defaults src/params.cc:40-109).
struct orc_params_t {
  double min_range;            // 3.0
  double max_range;            // 75.0
  int32_t lidar_type;          // 0 Velodyne, 1 Ouster
  int32_t scan_lines;          // 64
  int32_t scan_regions;        // 8
  int32_t edges_per_region;    // 10
  int64_t min_points_per_scan; // scan_regions*edges_per_region + 10 (params.cc:63)
  int64_t local_map_size;      // prev_frames, 5
  int32_t filter_local_map;    // false
  int32_t mapping;             // false
  // Switch for the one Ceres detail that cannot be verified here: 0 = on a
  // function-tolerance stop the candidate step is NOT applied (Ceres >= 1.12 behaviour).
  int32_t lm_apply_step_on_ftol;
  int32_t knn_mode;            // 0 brute force (checker), 1 kd-tree (baseline timing)
  // Eigen::Transform::rotation() at src/laser_odometry.cc:164,186,403,420.  1 = Eigen 3.3.x (the
  // README's platform, Ubuntu 20.04 = Eigen 3.3.7): computeRotationScaling(), i.e. the orthonormal
  // polar factor U V^T of linear() from a JacobiSVD, for every Mode.  0 = Eigen >= 3.4, where
  // rotation() of an Isometry is an alias of linear().  No Eigen source in this image: declared.
  int32_t pose_rotation_mode;
  int32_t pad_;
};

}  // extern "C"

namespace {

// Thread policy of the reference, for the timed CPU baseline only (results do not depend on it):
//   stencil loop   #pragma omp parallel for num_threads(ncores_), ncores_ = max(2, omp_get_max_threads() - 5)
//                  (src/feature_extractor.cc:29-34,194)
//   residual eval  ceres::Solver::Options::num_threads = nproc (src/laser_odometry.cc:216): Ceres evaluates the
//                  residual blocks in parallel; the linear solve stays serial
// 1 / 1 = everything serial (default).
int g_stencil_threads = 1;
int g_eval_threads = 1;

struct P4 { float x, y, z, i; };

// ------------------------------------------------------------------------------------------
// A1  FeatureExtractor::isValidPoint            src/feature_extractor.cc:84-102
// ------------------------------------------------------------------------------------------
inline bool is_valid_point(const orc_params_t& p, double x, double y, double z, double* dist) {
  bool valid = true;
  if (!std::isfinite(x) || !std::isfinite(y) || !std::isfinite(z)) valid = false;   // :89-93
  *dist = std::sqrt(x * x + y * y);                                                    // :96
  if (*dist > p.max_range || *dist < p.min_range) valid = false;                       // :97-99
  return valid;
}

// ------------------------------------------------------------------------------------------
// A2  FeatureExtractor::splitPointCloud         src/feature_extractor.cc:104-179
// Returns ring id or -1 for a Velodyne-type point (lidar_type 0).
// ------------------------------------------------------------------------------------------
inline int velodyne_ring(const orc_params_t& p, double z, double distance) {
  int scan_id = -1;
  double angle = std::atan(z / distance) * 180 / M_PI;                                 // :128
  if (p.scan_lines == 64) {                                                            // :130
    if (angle >= -8.83) scan_id = int((2 - angle) * 3.0 + 0.5);                        // :131-132
    else scan_id = p.scan_lines / 2 + int((-8.83 - angle) * 2.0 + 0.5);                // :134
    if (angle > 2 || angle < -24.33 || scan_id > 63 || scan_id < 0) return -1;         // :136-138
  } else if (p.scan_lines == 32) {                                                     // :139
    scan_id = int((angle + 92.0 / 3.0) * 3.0 / 4.0);                                   // :140
    if (scan_id > (p.scan_lines - 1) || scan_id < 0) return -1;                        // :141-143
  } else if (p.scan_lines == 16) {                                                     // :144
    scan_id = int((angle + 15) / 2 + 0.5);                                             // :145
    if (scan_id > (p.scan_lines - 1) || scan_id < 0) return -1;                        // :146-148
  } else {
    return -1;                                                                         // :149-151
  }
  return scan_id;
}

// rings[r] = source indices (into the input cloud) of ring r, in input order.
void split_point_cloud(const orc_params_t& p, const P4* pc, int64_t n, int height, int width,
                       std::vector<std::vector<int32_t>>& rings) {
  rings.assign(p.scan_lines, std::vector<int32_t>());                                 // :107-110
  if (p.lidar_type == 0) {                                                             // :113
    for (int64_t i = 0; i < n; i++) {                                                  // :115
      double x = pc[i].x, y = pc[i].y, z = pc[i].z;
      double distance;
      if (!is_valid_point(p, x, y, z, &distance)) continue;                            // :122-124
      int scan_id = velodyne_ring(p, z, distance);
      if (scan_id != -1) rings[scan_id].push_back((int32_t)i);                         // :154-156
    }
  } else if (p.lidar_type == 1) {                                                      // :158
    for (int row = 0; row < height; row++) {                                           // :160
      for (int col = 0; col < width; col++) {                                          // :161
        const P4& q = pc[(int64_t)row * width + col];                                  // at(col,row)
        double distance;
        if (!is_valid_point(p, q.x, q.y, q.z, &distance)) continue;                    // :168-170
        if (row < p.scan_lines) rings[row].push_back((int32_t)((int64_t)row * width + col));  // :173
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// A3  SmoothnessItem                             include/liodom/feature_extractor.h:44-60
// operator< sorts by descending smoothness (:57-59).  Ties are unspecified by std::sort in the
// reference; this restatement (and the GPU path) defines them as "lower point index first".
// ------------------------------------------------------------------------------------------
struct SmoothnessItem {
  int point_index = -1;
  double smoothness = -1.0;
};
inline bool smooth_less(const SmoothnessItem& a, const SmoothnessItem& b) {
  if (a.smoothness != b.smoothness) return a.smoothness > b.smoothness;
  return a.point_index < b.point_index;
}

struct EdgeOut {
  std::vector<P4> pts;
  std::vector<int32_t> ring, idx_in_ring, src;
};

// ------------------------------------------------------------------------------------------
// A5  FeatureExtractor::extractFeaturesFromRegion   src/feature_extractor.cc:256-313
// ------------------------------------------------------------------------------------------
void extract_from_region(const orc_params_t& p, const std::vector<P4>& ring_pts,
                         const std::vector<int32_t>& ring_src, int ring_id,
                         std::vector<SmoothnessItem>& smooths, std::vector<uint8_t>& picked,
                         EdgeOut& out) {
  std::sort(smooths.begin(), smooths.end(), smooth_less);                              // :261
  int picked_edges = 0;                                                                // :264
  for (size_t i = 0; i < smooths.size(); i++) {                                        // :265
    int point_index = smooths[i].point_index;
    if (!picked[point_index]) {                                                        // :268
      if (smooths[i].smoothness < 0.1 || picked_edges > p.edges_per_region) break;     // :270-272
      out.pts.push_back(ring_pts[point_index]);                                        // :275
      out.ring.push_back(ring_id);
      out.idx_in_ring.push_back(point_index);
      out.src.push_back(ring_src[point_index]);
      picked_edges++;                                                                  // :276
      picked[point_index] = 1;                                                         // :277
      for (int l = 1; l <= 5; l++) {                                                   // :280
        // float - float is a float subtraction; the result is then widened (:281-286)
        double dx = ring_pts[point_index + l].x - ring_pts[point_index + l - 1].x;
        double dy = ring_pts[point_index + l].y - ring_pts[point_index + l - 1].y;
        double dz = ring_pts[point_index + l].z - ring_pts[point_index + l - 1].z;
        if (dx * dx + dy * dy + dz * dz > 0.05) break;                                 // :289-291
        picked[point_index + l] = 1;                                                   // :293
      }
      for (int l = -1; l >= -5; l--) {                                                 // :296
        double dx = ring_pts[point_index + l].x - ring_pts[point_index + l + 1].x;     // :297-302, float subtraction
        double dy = ring_pts[point_index + l].y - ring_pts[point_index + l + 1].y;
        double dz = ring_pts[point_index + l].z - ring_pts[point_index + l + 1].z;
        if (dx * dx + dy * dy + dz * dz > 0.05) break;                                 // :305-307
        picked[point_index + l] = 1;                                                   // :309
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// A4  FeatureExtractor::extractFeatures          src/feature_extractor.cc:181-254
// curv_out (optional): per ring, the smoothness of every compacted point (NaN where undefined).
// ------------------------------------------------------------------------------------------
void extract_features(const orc_params_t& p, const P4* pc,
                      const std::vector<std::vector<int32_t>>& rings, EdgeOut& out,
                      std::vector<std::vector<double>>* curv_out) {
  if (curv_out) curv_out->assign(p.scan_lines, std::vector<double>());
  for (int i = 0; i < p.scan_lines; i++) {                                             // :186
    const std::vector<int32_t>& src = rings[i];
    const int64_t n = (int64_t)src.size();
    if (n < p.min_points_per_scan) continue;                                           // :188-190
    std::vector<P4> pts(n);
    for (int64_t k = 0; k < n; k++) pts[k] = pc[src[k]];
    std::vector<SmoothnessItem> smooths_aux(n);                                        // :193
    std::vector<uint8_t> picked(n, 0);  // picked_ is reset only for j in [5, n-5) (:230);
                                        // entries outside are written but never read.
#pragma omp parallel for num_threads(g_stencil_threads) if (g_stencil_threads > 1)           // :194
    for (int64_t j = 5; j < n - 5; j++) {                                              // :195
      // The operands are pcl::PointXYZI floats and `10 * x` is int * float, so each sum is a
      // FLOAT expression evaluated left to right (x86-64 SSE, FLT_EVAL_METHOD 0) and only the
      // result is widened by the `double diff_x =` initialisation (:196-228).
      double diff_x = pts[j - 5].x + pts[j - 4].x + pts[j - 3].x + pts[j - 2].x + pts[j - 1].x -
                      10 * pts[j].x + pts[j + 1].x + pts[j + 2].x + pts[j + 3].x + pts[j + 4].x +
                      pts[j + 5].x;                                                    // :196-206
      double diff_y = pts[j - 5].y + pts[j - 4].y + pts[j - 3].y + pts[j - 2].y + pts[j - 1].y -
                      10 * pts[j].y + pts[j + 1].y + pts[j + 2].y + pts[j + 3].y + pts[j + 4].y +
                      pts[j + 5].y;                                                    // :207-217
      double diff_z = pts[j - 5].z + pts[j - 4].z + pts[j - 3].z + pts[j - 2].z + pts[j - 1].z -
                      10 * pts[j].z + pts[j + 1].z + pts[j + 2].z + pts[j + 3].z + pts[j + 4].z +
                      pts[j + 5].z;                                                    // :218-228
      smooths_aux[j].point_index = (int)j;
      smooths_aux[j].smoothness = diff_x * diff_x + diff_y * diff_y + diff_z * diff_z; // :229
      picked[j] = 0;                                                                   // :230
    }
    if (curv_out) {
      (*curv_out)[i].assign(n, std::numeric_limits<double>::quiet_NaN());
      for (int64_t j = 5; j < n - 5; j++) (*curv_out)[i][j] = smooths_aux[j].smoothness;
    }
    std::vector<SmoothnessItem> smooths(smooths_aux.begin() + 5, smooths_aux.end() - 5);  // :235
    int total_points = (int)n - 10;                                                    // :238
    int sector_length = (int)(total_points / p.scan_regions);                          // :239
    for (int j = 0; j < p.scan_regions; j++) {                                         // :240
      int region_start = sector_length * j;                                            // :242
      int region_end = sector_length * (j + 1);                                        // :243
      if (j == p.scan_regions - 1) region_end = total_points;                          // :244-247
      std::vector<SmoothnessItem> sub(smooths.begin() + region_start, smooths.begin() + region_end);
      extract_from_region(p, pts, src, i, sub, picked, out);                           // :251
    }
  }
}

// ------------------------------------------------------------------------------------------
// Pose helpers: Eigen 3.3 semantics used at src/laser_odometry.cc:148-150,186-195,222-227.
// T is a 3x4 row-major [R | t] (the top rows of Eigen::Isometry3d::matrix()).
// ------------------------------------------------------------------------------------------
struct Iso { double m[12]; };
inline Iso iso_identity() { Iso I{}; I.m[0] = I.m[5] = I.m[10] = 1.0; return I; }
inline double R_(const Iso& T, int r, int c) { return T.m[r * 4 + c]; }
inline Iso iso_mul(const Iso& A, const Iso& B) {
  Iso C{};
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++)
      C.m[r * 4 + c] = R_(A, r, 0) * R_(B, 0, c) + R_(A, r, 1) * R_(B, 1, c) + R_(A, r, 2) * R_(B, 2, c);
    C.m[r * 4 + 3] = R_(A, r, 0) * B.m[3] + R_(A, r, 1) * B.m[7] + R_(A, r, 2) * B.m[11] + A.m[r * 4 + 3];
  }
  return C;
}
// Isometry inverse: R^T, -R^T t (Eigen Transform::inverse(Isometry)).
inline Iso iso_inverse(const Iso& A) {
  Iso C{};
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) C.m[r * 4 + c] = R_(A, c, r);
  for (int r = 0; r < 3; r++)
    C.m[r * 4 + 3] = -(C.m[r * 4 + 0] * A.m[3] + C.m[r * 4 + 1] * A.m[7] + C.m[r * 4 + 2] * A.m[11]);
  return C;
}
// Eigen::Quaterniond(Matrix3d): quaternionbase_assign_impl<Other,3,3> (Shepperd's branches).
// q = [x, y, z, w].
inline void quat_from_rot(const Iso& T, double q[4]) {
  double t = R_(T, 0, 0) + R_(T, 1, 1) + R_(T, 2, 2);
  if (t > 0.0) {
    t = std::sqrt(t + 1.0);
    q[3] = 0.5 * t;
    t = 0.5 / t;
    q[0] = (R_(T, 2, 1) - R_(T, 1, 2)) * t;
    q[1] = (R_(T, 0, 2) - R_(T, 2, 0)) * t;
    q[2] = (R_(T, 1, 0) - R_(T, 0, 1)) * t;
  } else {
    int i = 0;
    if (R_(T, 1, 1) > R_(T, 0, 0)) i = 1;
    if (R_(T, 2, 2) > R_(T, i, i)) i = 2;
    int j = (i + 1) % 3, k = (j + 1) % 3;
    t = std::sqrt(R_(T, i, i) - R_(T, j, j) - R_(T, k, k) + 1.0);
    q[i] = 0.5 * t;
    t = 0.5 / t;
    q[3] = (R_(T, k, j) - R_(T, j, k)) * t;
    q[j] = (R_(T, j, i) + R_(T, i, j)) * t;
    q[k] = (R_(T, k, i) + R_(T, i, k)) * t;
  }
}
// Eigen::Quaterniond::toRotationMatrix (no normalisation).
inline Iso iso_from_qt(const double q[4], const double t[3]) {
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w;
  const double txx = tx * x, txy = ty * x, txz = tz * x;
  const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
  Iso T{};
  T.m[0] = 1 - (tyy + tzz); T.m[1] = txy - twz;       T.m[2] = txz + twy;        T.m[3] = t[0];
  T.m[4] = txy + twz;       T.m[5] = 1 - (txx + tzz); T.m[6] = tyz - twx;        T.m[7] = t[1];
  T.m[8] = txz - twy;       T.m[9] = tyz + twx;       T.m[10] = 1 - (txx + tyy); T.m[11] = t[2];
  return T;
}

// Eigen 3.3 Transform::rotation() -> computeRotationScaling(&R, 0) (Eigen/src/Geometry/Transform.h):
//   JacobiSVD svd(linear(), ComputeFullU | ComputeFullV);
//   x = (U * V^T).determinant();  m = U;  m.col(0) /= x;  R = m * V^T
// i.e. the orthonormal polar factor of linear(), made a proper rotation.  Restated with a one-sided
// (Hestenes) Jacobi SVD — only + - * / sqrt; Eigen's two-sided sweep order differs, but the polar
// factor of a nonsingular matrix is unique, so the results agree to rounding (~1e-16).
inline void svd3(const double A[9], double U[9], double S[3], double V[9]) {
  double W[9];
  for (int i = 0; i < 9; i++) { W[i] = A[i]; V[i] = (i % 4 == 0) ? 1.0 : 0.0; }
  for (int sweep = 0; sweep < 30; sweep++) {
    bool rotated = false;
    for (int p = 0; p < 2; p++) {
      for (int q = p + 1; q < 3; q++) {
        double alpha = 0, beta = 0, gamma = 0;
        for (int i = 0; i < 3; i++) { alpha += W[i * 3 + p] * W[i * 3 + p]; beta += W[i * 3 + q] * W[i * 3 + q]; gamma += W[i * 3 + p] * W[i * 3 + q]; }
        if (std::fabs(gamma) <= 1e-17 * std::sqrt(alpha * beta) || gamma == 0.0) continue;
        rotated = true;
        const double zeta = (beta - alpha) / (2.0 * gamma);
        double t = 1.0 / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
        if (zeta < 0.0) t = -t;
        const double c = 1.0 / std::sqrt(1.0 + t * t), sn = c * t;
        for (int i = 0; i < 3; i++) {
          const double wp = W[i * 3 + p], wq = W[i * 3 + q];
          W[i * 3 + p] = c * wp - sn * wq; W[i * 3 + q] = sn * wp + c * wq;
          const double vp = V[i * 3 + p], vq = V[i * 3 + q];
          V[i * 3 + p] = c * vp - sn * vq; V[i * 3 + q] = sn * vp + c * vq;
        }
      }
    }
    if (!rotated) break;
  }
  for (int j = 0; j < 3; j++) {
    double n = 0;
    for (int i = 0; i < 3; i++) n += W[i * 3 + j] * W[i * 3 + j];
    n = std::sqrt(n);
    S[j] = n;
    for (int i = 0; i < 3; i++) U[i * 3 + j] = W[i * 3 + j] / n;
  }
}
inline double det3(const double M[9]) {
  return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) + M[2] * (M[3] * M[7] - M[4] * M[6]);
}
// T with its linear part replaced by what Transform::rotation() returns (mode 1), or T itself (mode 0).
inline Iso rotation_of(const Iso& T, int mode) {
  if (mode == 0) return T;
  double A[9], U[9], S[3], V[9], UVt[9];
  for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) A[r * 3 + c] = R_(T, r, c);
  svd3(A, U, S, V);
  for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) UVt[r * 3 + c] = U[r * 3 + 0] * V[c * 3 + 0] + U[r * 3 + 1] * V[c * 3 + 1] + U[r * 3 + 2] * V[c * 3 + 2];
  const double x = det3(UVt);                      // +-1
  for (int r = 0; r < 3; r++) U[r * 3 + 0] /= x;   // m.col(0) /= x
  Iso R = T;
  for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) R.m[r * 4 + c] = U[r * 3 + 0] * V[c * 3 + 0] + U[r * 3 + 1] * V[c * 3 + 1] + U[r * 3 + 2] * V[c * 3 + 2];
  return R;
}

// ------------------------------------------------------------------------------------------
// IMU roll / pitch override of the predicted pose        src/laser_odometry.cc:152-183
// The reference goes through ROS tf's LinearMath (tf::Quaternion, tf::Matrix3x3: getRPY, setRPY,
// getRotation).  tf is a third-party dependency that is NOT in /root/reference and is not version
// pinned (package.xml lists `tf`; ROS Noetic ships geometry 1.13.x).  The three functions below
// restate the published algorithms of tf/LinearMath/Matrix3x3.h (Bullet's btMatrix3x3, tfScalar =
// double): parity unpinned, cross-checked against scipy's fixed-axis Euler conversion in
// tests/test_oracle_odometry.py.
// ------------------------------------------------------------------------------------------
struct Mat3 { double e[3][3]; };
// tf::Matrix3x3::setRotation(q): not Eigen's toRotationMatrix — it divides by |q|^2.
inline Mat3 tf_matrix_from_quat(const double q[4]) {
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  const double d = x * x + y * y + z * z + w * w;
  const double sc = 2.0 / d;
  const double xs = x * sc, ys = y * sc, zs = z * sc;
  const double wx = w * xs, wy = w * ys, wz = w * zs;
  const double xx = x * xs, xy = x * ys, xz = x * zs;
  const double yy = y * ys, yz = y * zs, zz = z * zs;
  Mat3 m;
  m.e[0][0] = 1.0 - (yy + zz); m.e[0][1] = xy - wz;         m.e[0][2] = xz + wy;
  m.e[1][0] = xy + wz;         m.e[1][1] = 1.0 - (xx + zz); m.e[1][2] = yz - wx;
  m.e[2][0] = xz - wy;         m.e[2][1] = yz + wx;         m.e[2][2] = 1.0 - (xx + yy);
  return m;
}
// tf::Matrix3x3::getRPY = getEulerYPR, solution 1 (including its gimbal-lock branch)
inline void tf_get_rpy(const Mat3& m, double* roll, double* pitch, double* yaw) {
  if (std::fabs(m.e[2][0]) >= 1) {
    *yaw = 0;
    const double delta = std::atan2(m.e[2][1], m.e[2][2]);
    if (m.e[2][0] < 0) { *pitch = M_PI / 2.0; *roll = delta; }
    else { *pitch = -M_PI / 2.0; *roll = delta; }
  } else {
    *pitch = -std::asin(m.e[2][0]);
    const double cp = std::cos(*pitch);
    *roll = std::atan2(m.e[2][1] / cp, m.e[2][2] / cp);
    *yaw = std::atan2(m.e[1][0] / cp, m.e[0][0] / cp);
  }
}
// tf::Matrix3x3::setRPY(roll, pitch, yaw) = setEulerYPR(yaw, pitch, roll)
inline Mat3 tf_set_rpy(double roll, double pitch, double yaw) {
  const double ci = std::cos(roll), cj = std::cos(pitch), ch = std::cos(yaw);
  const double si = std::sin(roll), sj = std::sin(pitch), sh = std::sin(yaw);
  const double cc = ci * ch, cs = ci * sh, sc = si * ch, ss = si * sh;
  Mat3 m;
  m.e[0][0] = cj * ch; m.e[0][1] = sj * sc - cs; m.e[0][2] = sj * cc + ss;
  m.e[1][0] = cj * sh; m.e[1][1] = sj * ss + cc; m.e[1][2] = sj * cs - sc;
  m.e[2][0] = -sj;     m.e[2][1] = cj * si;      m.e[2][2] = cj * ci;
  return m;
}
// tf::Matrix3x3::getRotation(q), q = [x y z w]
inline void tf_quat_from_matrix(const Mat3& m, double q[4]) {
  const double trace = m.e[0][0] + m.e[1][1] + m.e[2][2];
  if (trace > 0.0) {
    double sq = std::sqrt(trace + 1.0);
    q[3] = sq * 0.5;
    sq = 0.5 / sq;
    q[0] = (m.e[2][1] - m.e[1][2]) * sq;
    q[1] = (m.e[0][2] - m.e[2][0]) * sq;
    q[2] = (m.e[1][0] - m.e[0][1]) * sq;
  } else {
    const int i = m.e[0][0] < m.e[1][1] ? (m.e[1][1] < m.e[2][2] ? 2 : 1) : (m.e[0][0] < m.e[2][2] ? 2 : 0);
    const int j = (i + 1) % 3, k = (i + 2) % 3;
    double sq = std::sqrt(m.e[i][i] - m.e[j][j] - m.e[k][k] + 1.0);
    q[i] = sq * 0.5;
    sq = 0.5 / sq;
    q[3] = (m.e[k][j] - m.e[j][k]) * sq;
    q[j] = (m.e[j][i] + m.e[i][j]) * sq;
    q[k] = (m.e[k][i] + m.e[i][k]) * sq;
  }
}
// src/laser_odometry.cc:152-183, steps 1-5 as numbered there
inline Iso imu_override(const Iso& odom, const double imu_q[4], const Iso& laser_to_base, int rotation_mode) {
  double imu_roll, imu_pitch, imu_yaw;
  tf_get_rpy(tf_matrix_from_quat(imu_q), &imu_roll, &imu_pitch, &imu_yaw);            // :155-161
  Iso odom_bl = iso_mul(odom, laser_to_base);                                         // :164
  double q_bl[4];
  quat_from_rot(rotation_of(odom_bl, rotation_mode), q_bl);                           // :165 Quaterniond(odom_bl.rotation())
  double bl_roll, bl_pitch, bl_yaw;
  tf_get_rpy(tf_matrix_from_quat(q_bl), &bl_roll, &bl_pitch, &bl_yaw);                // :166-169
  const Mat3 m = tf_set_rpy(imu_roll, imu_pitch, bl_yaw);                             // :174
  double q_new[4];
  tf_quat_from_matrix(m, q_new);                                                      // :177
  const double t_bl[3] = {odom_bl.m[3], odom_bl.m[7], odom_bl.m[11]};
  odom_bl = iso_from_qt(q_new, t_bl);                                                 // :178-179 (Eigen toRotationMatrix)
  return iso_mul(odom_bl, iso_inverse(laser_to_base));                                // :182
}

// pcl::transformPointCloud with a double matrix (src/laser_odometry.cc:232,308): PCL 1.10
// detail::Transformer<double>::se3 — each coordinate in FP64, left to right, cast to float;
// intensity copied.
inline P4 transform_point(const Iso& T, const P4& p) {
  const double x = p.x, y = p.y, z = p.z;
  P4 o;
  o.x = (float)(T.m[0] * x + T.m[1] * y + T.m[2] * z + T.m[3]);
  o.y = (float)(T.m[4] * x + T.m[5] * y + T.m[6] * z + T.m[7]);
  o.z = (float)(T.m[8] * x + T.m[9] * y + T.m[10] * z + T.m[11]);
  o.i = p.i;
  return o;
}

// ------------------------------------------------------------------------------------------
// 5-NN, FLANN L2_Simple<float> arithmetic (src/laser_odometry.cc:318-323):
// float differences, float accumulate in x->y->z order, results ascending by distance.
// Ties: lower map index first (declared; FLANN's tie order is an implementation detail).
// ------------------------------------------------------------------------------------------
inline float sqdist_f(const P4& a, const P4& b) {
  float dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z;
  float r = dx * dx;
  r = r + dy * dy;
  r = r + dz * dz;
  return r;
}

struct Knn5 { int idx[5]; float d[5]; int n; };

inline void knn_insert(Knn5& k, float d, int idx) {
  // keeps (d, idx) ascending lexicographically, at most 5 entries
  int pos = k.n;
  if (k.n == 5) {
    if (!(d < k.d[4] || (d == k.d[4] && idx < k.idx[4]))) return;
    pos = 4;
  } else {
    k.n++;
  }
  while (pos > 0 && (d < k.d[pos - 1] || (d == k.d[pos - 1] && idx < k.idx[pos - 1]))) {
    k.d[pos] = k.d[pos - 1]; k.idx[pos] = k.idx[pos - 1];
    pos--;
  }
  k.d[pos] = d; k.idx[pos] = idx;
}

void knn5_brute(const std::vector<P4>& map, const P4& q, Knn5& out) {
  out.n = 0;
  for (size_t m = 0; m < map.size(); m++) knn_insert(out, sqdist_f(q, map[m]), (int)m);
}

// kd-tree with the reference's cost structure (built per call to addEdgeConstraints, then E
// serial queries).  Median split on the widest dimension, leaves of <= 15 points (FLANN
// KDTreeSingleIndex leaf_max_size as set by pcl::KdTreeFLANN).  Exact search, same distance
// arithmetic as the brute-force path, so results are identical (ties aside).
struct KdTree {
  struct Node { int lo, hi; int dim; float split; int left, right; float bmin[3], bmax[3]; };
  std::vector<Node> nodes;
  std::vector<int> perm;
  const std::vector<P4>* pts = nullptr;
  static float coord(const P4& p, int d) { return d == 0 ? p.x : (d == 1 ? p.y : p.z); }
  int build_rec(int lo, int hi) {
    Node nd; nd.lo = lo; nd.hi = hi; nd.left = nd.right = -1; nd.dim = 0; nd.split = 0;
    for (int d = 0; d < 3; d++) { nd.bmin[d] = FLT_MAX; nd.bmax[d] = -FLT_MAX; }
    for (int i = lo; i < hi; i++)
      for (int d = 0; d < 3; d++) {
        float c = coord((*pts)[perm[i]], d);
        nd.bmin[d] = std::min(nd.bmin[d], c); nd.bmax[d] = std::max(nd.bmax[d], c);
      }
    int id = (int)nodes.size();
    nodes.push_back(nd);
    if (hi - lo > 15) {
      int dim = 0; float best = -1;
      for (int d = 0; d < 3; d++) { float e = nd.bmax[d] - nd.bmin[d]; if (e > best) { best = e; dim = d; } }
      int mid = (lo + hi) / 2;
      std::nth_element(perm.begin() + lo, perm.begin() + mid, perm.begin() + hi,
                       [&](int a, int b) { return coord((*pts)[a], dim) < coord((*pts)[b], dim); });
      nodes[id].dim = dim;
      nodes[id].split = coord((*pts)[perm[mid]], dim);
      int l = build_rec(lo, mid);
      int r = build_rec(mid, hi);
      nodes[id].left = l; nodes[id].right = r;
    }
    return id;
  }
  void build(const std::vector<P4>& p) {
    pts = &p; nodes.clear(); perm.resize(p.size());
    for (size_t i = 0; i < p.size(); i++) perm[i] = (int)i;
    if (!p.empty()) { nodes.reserve(p.size() / 4 + 16); build_rec(0, (int)p.size()); }
  }
  // lower bound of the float squared distance from q to a node's box (conservative: computed
  // in double and compared with a small slack so float rounding can never prune a true result)
  static double box_dist(const Node& nd, const P4& q) {
    double s = 0;
    for (int d = 0; d < 3; d++) {
      double c = coord(q, d), e = 0;
      if (c < nd.bmin[d]) e = nd.bmin[d] - c; else if (c > nd.bmax[d]) e = c - nd.bmax[d];
      s += e * e;
    }
    return s;
  }
  void search_rec(int id, const P4& q, Knn5& k) const {
    const Node& nd = nodes[id];
    if (k.n == 5 && box_dist(nd, q) * (1.0 - 1e-5) > (double)k.d[4]) return;
    if (nd.left < 0) {
      for (int i = nd.lo; i < nd.hi; i++) knn_insert(k, sqdist_f(q, (*pts)[perm[i]]), perm[i]);
      return;
    }
    bool left_first = coord(q, nd.dim) < nd.split;
    search_rec(left_first ? nd.left : nd.right, q, k);
    search_rec(left_first ? nd.right : nd.left, q, k);
  }
  void search(const P4& q, Knn5& k) const { k.n = 0; if (!nodes.empty()) search_rec(0, q, k); }
};

// ------------------------------------------------------------------------------------------
// Eigenvalues of a symmetric 3x3 (src/laser_odometry.cc:342, Eigen::SelfAdjointEigenSolver;
// only eigenvalues are used, :344).  Restated as 8 fixed sweeps of cyclic Jacobi using only
// + - * / sqrt, so a GPU implementation of the same sequence is bit-identical.
// a = {a00, a01, a02, a11, a12, a22}; ev ascending.
// ------------------------------------------------------------------------------------------
inline void jacobi_rot(double& app, double& aqq, double& apq, double& arp, double& arq) {
  // converged pair: rotating further cannot change the eigenvalues at double precision (the
  // same test, bit for bit, in the oracle and in the GPU path)
  if (std::fabs(apq) <= 1e-20 * (std::fabs(app) + std::fabs(aqq))) return;
  double theta = (aqq - app) / (2.0 * apq);
  double at = std::fabs(theta);
  double t = 1.0 / (at + std::sqrt(theta * theta + 1.0));
  if (theta < 0.0) t = -t;
  double c = 1.0 / std::sqrt(t * t + 1.0);
  double s = t * c;
  app = app - t * apq;
  aqq = aqq + t * apq;
  apq = 0.0;
  double nrp = c * arp - s * arq;
  double nrq = s * arp + c * arq;
  arp = nrp; arq = nrq;
}
void eig3_sym(const double a[6], double ev[3]) {
  double a00 = a[0], a01 = a[1], a02 = a[2], a11 = a[3], a12 = a[4], a22 = a[5];
  for (int sweep = 0; sweep < 8; sweep++) {
    jacobi_rot(a00, a11, a01, a02, a12);   // (p,q)=(0,1), r=2: arp=a02, arq=a12
    jacobi_rot(a00, a22, a02, a01, a12);   // (0,2), r=1: arp=a01, arq=a21=a12
    jacobi_rot(a11, a22, a12, a01, a02);   // (1,2), r=0: arp=a10=a01, arq=a20=a02
  }
  double e0 = a00, e1 = a11, e2 = a22, s;
  if (e0 > e1) { s = e0; e0 = e1; e1 = s; }
  if (e1 > e2) { s = e1; e1 = e2; e2 = s; }
  if (e0 > e1) { s = e0; e0 = e1; e1 = s; }
  ev[0] = e0; ev[1] = e1; ev[2] = e2;
}

// ------------------------------------------------------------------------------------------
// Forward-mode dual numbers: a literal restatement of ceres::Jet<double,7> as used by
// AutoDiffCostFunction<Point2LineFactor,3,4,3> (include/liodom/factors.hpp:112).
// ------------------------------------------------------------------------------------------
struct Jet {
  double a; double v[7];
  Jet() : a(0) { for (int i = 0; i < 7; i++) v[i] = 0; }
  Jet(double s) : a(s) { for (int i = 0; i < 7; i++) v[i] = 0; }  // NOLINT
  Jet(double s, int k) : a(s) { for (int i = 0; i < 7; i++) v[i] = 0; v[k] = 1.0; }
};
inline Jet operator+(const Jet& f, const Jet& g) { Jet h; h.a = f.a + g.a; for (int i = 0; i < 7; i++) h.v[i] = f.v[i] + g.v[i]; return h; }
inline Jet operator-(const Jet& f, const Jet& g) { Jet h; h.a = f.a - g.a; for (int i = 0; i < 7; i++) h.v[i] = f.v[i] - g.v[i]; return h; }
inline Jet operator-(const Jet& f) { Jet h; h.a = -f.a; for (int i = 0; i < 7; i++) h.v[i] = -f.v[i]; return h; }
inline Jet operator*(const Jet& f, const Jet& g) { Jet h; h.a = f.a * g.a; for (int i = 0; i < 7; i++) h.v[i] = f.a * g.v[i] + f.v[i] * g.a; return h; }
inline Jet operator/(const Jet& f, const Jet& g) {
  // ceres/jet.h: g_inverse = 1/g.a; f_a_by_g_a = f.a*g_inverse; v = (f.v - f_a_by_g_a*g.v)*g_inverse
  Jet h; const double gi = 1.0 / g.a; const double fg = f.a * gi; h.a = fg;
  for (int i = 0; i < 7; i++) h.v[i] = (f.v[i] - fg * g.v[i]) * gi;
  return h;
}
inline Jet jsqrt(const Jet& f) { Jet h; h.a = std::sqrt(f.a); const double t = 1.0 / (2.0 * h.a); for (int i = 0; i < 7; i++) h.v[i] = t * f.v[i]; return h; }
inline Jet jsin(const Jet& f) { Jet h; h.a = std::sin(f.a); const double c = std::cos(f.a); for (int i = 0; i < 7; i++) h.v[i] = c * f.v[i]; return h; }
inline Jet jacos(const Jet& f) { Jet h; h.a = std::acos(f.a); const double t = -1.0 / std::sqrt(1.0 - f.a * f.a); for (int i = 0; i < 7; i++) h.v[i] = t * f.v[i]; return h; }
inline Jet jabs(const Jet& f) { return f.a < 0.0 ? -f : f; }

struct JQuat { Jet w, x, y, z; };

// Eigen::QuaternionBase::slerp (Eigen 3.3 Geometry/Quaternion.h) on Jets.
inline JQuat jslerp(const JQuat& self, const Jet& t, const JQuat& other) {
  const double one = 1.0 - std::numeric_limits<double>::epsilon();
  Jet d = self.w * other.w + self.x * other.x + self.y * other.y + self.z * other.z;
  Jet absD = jabs(d);
  Jet scale0, scale1;
  if (absD.a >= one) {
    scale0 = Jet(1.0) - t;
    scale1 = t;
  } else {
    Jet theta = jacos(absD);
    Jet sinTheta = jsin(theta);
    scale0 = jsin((Jet(1.0) - t) * theta) / sinTheta;
    scale1 = jsin(t * theta) / sinTheta;
  }
  if (d.a < 0.0) scale1 = -scale1;
  JQuat r;
  r.w = scale0 * self.w + scale1 * other.w;
  r.x = scale0 * self.x + scale1 * other.x;
  r.y = scale0 * self.y + scale1 * other.y;
  r.z = scale0 * self.z + scale1 * other.z;
  return r;
}

struct Corr { double p[3], a[3], b[3]; };

// ------------------------------------------------------------------------------------------
// A10  Point2LineFactor::operator()             include/liodom/factors.hpp:71-105
// Evaluates residual[3] and the 3x7 global Jacobian (d/d(qx,qy,qz,qw,tx,ty,tz)).
// ------------------------------------------------------------------------------------------
inline void point2line_jets(const Corr& c, const double q[4], const double t[3], double min_d,
                            double max_d, Jet res[3]) {
  Jet qj[4] = {Jet(q[0], 0), Jet(q[1], 1), Jet(q[2], 2), Jet(q[3], 3)};
  Jet tj[3] = {Jet(t[0], 4), Jet(t[1], 5), Jet(t[2], 6)};
  Jet cp[3] = {Jet(c.p[0]), Jet(c.p[1]), Jet(c.p[2])};                                 // :73
  Jet lpa[3] = {Jet(c.a[0]), Jet(c.a[1]), Jet(c.a[2])};                                // :74
  Jet lpb[3] = {Jet(c.b[0]), Jet(c.b[1]), Jet(c.b[2])};                                // :75
  JQuat q_last_curr{qj[3], qj[0], qj[1], qj[2]};                                       // :77
  JQuat q_identity{Jet(1.0), Jet(0.0), Jet(0.0), Jet(0.0)};                            // :78
  q_last_curr = jslerp(q_identity, Jet(1.0), q_last_curr);                             // :79
  Jet tl[3] = {Jet(1.0) * tj[0], Jet(1.0) * tj[1], Jet(1.0) * tj[2]};                  // :80
  // lp = q * cp + t  (:82-83); Eigen _transformVector: uv = vec x v; uv += uv;
  // v + w*uv + vec x uv
  const Jet& qx = q_last_curr.x; const Jet& qy = q_last_curr.y; const Jet& qz = q_last_curr.z;
  Jet uv[3] = {qy * cp[2] - qz * cp[1], qz * cp[0] - qx * cp[2], qx * cp[1] - qy * cp[0]};
  uv[0] = uv[0] + uv[0]; uv[1] = uv[1] + uv[1]; uv[2] = uv[2] + uv[2];
  Jet cr[3] = {qy * uv[2] - qz * uv[1], qz * uv[0] - qx * uv[2], qx * uv[1] - qy * uv[0]};
  Jet lp[3];
  for (int i = 0; i < 3; i++) lp[i] = (cp[i] + q_last_curr.w * uv[i] + cr[i]) + tl[i];
  Jet u[3] = {lp[0] - lpa[0], lp[1] - lpa[1], lp[2] - lpa[2]};
  Jet w_[3] = {lp[0] - lpb[0], lp[1] - lpb[1], lp[2] - lpb[2]};
  Jet nu[3] = {u[1] * w_[2] - u[2] * w_[1], u[2] * w_[0] - u[0] * w_[2], u[0] * w_[1] - u[1] * w_[0]};  // :85
  Jet de[3] = {lpa[0] - lpb[0], lpa[1] - lpb[1], lpa[2] - lpb[2]};                     // :86
  Jet cpl[3] = {Jet(c.p[0]) - tj[0], Jet(c.p[1]) - tj[1], Jet(c.p[2]) - tj[2]};        // :88
  Jet d = jsqrt(cpl[0] * cpl[0] + cpl[1] * cpl[1]);                                    // :90-92
  d = (d - Jet(min_d)) / (Jet(max_d) - Jet(min_d));                                    // :93
  Jet w = Jet(1.01) - d;                                                               // :97
  Jet den = jsqrt(de[0] * de[0] + de[1] * de[1] + de[2] * de[2]);                      // de.norm()
  res[0] = w * (nu[0] / den);                                                          // :99
  res[1] = w * (nu[1] / den);                                                          // :100
  res[2] = w * (nu[2] / den);                                                          // :101
}

// ceres::EigenQuaternionParameterization (Ceres <= 2.1 local_parameterization.cc), x = [x y z w].
inline void quat_plus(const double x[4], const double delta[3], double out[4]) {
  const double norm_delta = std::sqrt(delta[0] * delta[0] + delta[1] * delta[1] + delta[2] * delta[2]);
  if (norm_delta > 0.0) {
    const double s = std::sin(norm_delta) / norm_delta;
    const double tw = std::cos(norm_delta), tx = s * delta[0], ty = s * delta[1], tz = s * delta[2];
    // Eigen quaternion product tmp * x
    const double aw = tw, ax = tx, ay = ty, az = tz, bw = x[3], bx = x[0], by = x[1], bz = x[2];
    out[3] = aw * bw - ax * bx - ay * by - az * bz;
    out[0] = aw * bx + ax * bw + ay * bz - az * by;
    out[1] = aw * by + ay * bw + az * bx - ax * bz;
    out[2] = aw * bz + az * bw + ax * by - ay * bx;
  } else {
    out[0] = x[0]; out[1] = x[1]; out[2] = x[2]; out[3] = x[3];
  }
}
inline void quat_plus_jacobian(const double x[4], double J[12]) {  // 4x3 row-major
  J[0] = x[3];  J[1] = x[2];   J[2] = -x[1];
  J[3] = -x[2]; J[4] = x[3];   J[5] = x[0];
  J[6] = x[1];  J[7] = -x[0];  J[8] = x[3];
  J[9] = -x[0]; J[10] = -x[1]; J[11] = -x[2];
}

// One residual block as Ceres' ResidualBlock::Evaluate sees it: raw residual + local (tangent)
// Jacobian 3x6 (q block first, then t; src/laser_odometry.cc:205-206,360), Huber(0.2) loss
// (:201) applied through Corrector (rho'' <= 0 -> plain sqrt(rho') scaling).
// Returns false if anything is non-finite (Ceres rejects such evaluations).
inline bool eval_block(const Corr& c, const double q[4], const double t[3], double min_d,
                       double max_d, bool want_jac, double* cost, double r[3], double J[18]) {
  Jet res[3];
  point2line_jets(c, q, t, min_d, max_d, res);
  double s = 0;
  for (int i = 0; i < 3; i++) { r[i] = res[i].a; s += r[i] * r[i]; }
  if (!std::isfinite(s)) return false;
  // HuberLoss(a = 0.2): b = a^2
  const double a = 0.2, b = a * a;
  double rho0, rho1;
  if (s > b) { const double rr = std::sqrt(s); rho0 = 2.0 * a * rr - b; rho1 = std::max(std::numeric_limits<double>::min(), a / rr); }
  else { rho0 = s; rho1 = 1.0; }
  *cost = 0.5 * rho0;
  if (!want_jac) return true;
  double P[12];
  quat_plus_jacobian(q, P);
  for (int i = 0; i < 3; i++) {
    for (int k = 0; k < 3; k++) {
      double acc = 0;
      for (int m = 0; m < 4; m++) acc += res[i].v[m] * P[m * 3 + k];
      J[i * 6 + k] = acc;
    }
    for (int k = 0; k < 3; k++) J[i * 6 + 3 + k] = res[i].v[4 + k];
  }
  const double sq = std::sqrt(rho1);
  for (int i = 0; i < 18; i++) { J[i] *= sq; if (!std::isfinite(J[i])) return false; }
  for (int i = 0; i < 3; i++) r[i] *= sq;
  return true;
}

// Householder QR least squares: minimise ||A y - b|| for A (m x 6, row-major), m >= 6.
// Stands in for Ceres' DENSE_QR (src/laser_odometry.cc:213) on the stacked system [J; D].
// Fault injection (tests only): the next g_fail_linear_solves calls report LINEAR_SOLVER_FAILURE, which
// TrustRegionMinimizer treats as an invalid step.
static int g_fail_linear_solves = 0;
bool qr_solve6(std::vector<double>& A, std::vector<double>& b, int m, double y[6]) {
  if (g_fail_linear_solves > 0) { g_fail_linear_solves--; return false; }
  const int n = 6;
  for (int k = 0; k < n; k++) {
    double norm = 0;
    for (int i = k; i < m; i++) norm += A[i * n + k] * A[i * n + k];
    norm = std::sqrt(norm);
    if (norm == 0.0) return false;
    double alpha = A[k * n + k] > 0 ? -norm : norm;
    std::vector<double> v(m - k);
    for (int i = k; i < m; i++) v[i - k] = A[i * n + k];
    v[0] -= alpha;
    double vnorm2 = 0;
    for (double e : v) vnorm2 += e * e;
    if (vnorm2 == 0.0) continue;
    for (int j = k; j < n; j++) {
      double dot = 0;
      for (int i = k; i < m; i++) dot += v[i - k] * A[i * n + j];
      double f = 2.0 * dot / vnorm2;
      for (int i = k; i < m; i++) A[i * n + j] -= f * v[i - k];
    }
    double dot = 0;
    for (int i = k; i < m; i++) dot += v[i - k] * b[i];
    double f = 2.0 * dot / vnorm2;
    for (int i = k; i < m; i++) b[i] -= f * v[i - k];
  }
  for (int k = n - 1; k >= 0; k--) {
    double s = b[k];
    for (int j = k + 1; j < n; j++) s -= A[k * n + j] * y[j];
    if (A[k * n + k] == 0.0) return false;
    y[k] = s / A[k * n + k];
    if (!std::isfinite(y[k])) return false;
  }
  return true;
}

struct LmTrace {
  int iterations = 0;          // trust-region iterations executed (<= 4)
  int accepted = 0;
  int termination = 0;         // 0 max-iter, 1 param tol, 2 function tol, 3 gradient tol, 4 no residuals, 5 eval failure, 6 radius, 7 invalid steps
  double initial_cost = 0, final_cost = 0;
  double cost[5] = {0, 0, 0, 0, 0};     // candidate cost per iteration (1..4) ; [0] = initial
  double radius[5] = {0, 0, 0, 0, 0};   // radius used at iteration i
  int step_ok[5] = {0, 0, 0, 0, 0};
};

// ------------------------------------------------------------------------------------------
// A11  ceres::Solve with the options of src/laser_odometry.cc:212-218 (trust region, LM,
// DENSE_QR, max_num_iterations = 4, Jacobi scaling on, monotonic steps).  Follows Ceres 1.14
// trust_region_minimizer.cc / levenberg_marquardt_strategy.cc.
// ------------------------------------------------------------------------------------------
void lm_solve(const std::vector<Corr>& blocks, double q[4], double t[3], double min_d,
              double max_d, int apply_on_ftol, LmTrace* tr) {
  LmTrace local; LmTrace& T = tr ? *tr : local;
  T = LmTrace();
  const int C = (int)blocks.size();
  if (C == 0) { T.termination = 4; return; }
  const int m = 3 * C;
  std::vector<double> r(m), J((size_t)m * 6);
  double xq[4] = {q[0], q[1], q[2], q[3]}, xt[3] = {t[0], t[1], t[2]};
  std::vector<double> blk_cost(C), blk_r((size_t)3 * C), blk_J((size_t)18 * C);
  std::vector<char> blk_ok(C);
  auto evaluate = [&](const double* eq, const double* et, bool jac, double* cost, std::vector<double>& rr, std::vector<double>& JJ) -> bool {
    // blocks in parallel (Ceres' evaluator threads), totals summed in block order: same bits for any thread count
#pragma omp parallel for num_threads(g_eval_threads) if (g_eval_threads > 1) schedule(static)
    for (int i = 0; i < C; i++)
      blk_ok[i] = eval_block(blocks[i], eq, et, min_d, max_d, jac, &blk_cost[i], &blk_r[(size_t)3 * i], &blk_J[(size_t)18 * i]) ? 1 : 0;
    double total = 0;
    for (int i = 0; i < C; i++) {
      if (!blk_ok[i]) return false;
      total += blk_cost[i];
      if (jac) {
        for (int k = 0; k < 3; k++) rr[3 * i + k] = blk_r[(size_t)3 * i + k];
        for (int k = 0; k < 18; k++) JJ[(size_t)(3 * i) * 6 + k] = blk_J[(size_t)18 * i + k];
      }
    }
    *cost = total;
    return true;
  };
  double x_cost;
  if (!evaluate(xq, xt, true, &x_cost, r, J)) { T.termination = 5; return; }
  T.initial_cost = x_cost; T.cost[0] = x_cost; T.final_cost = x_cost;
  auto norm7 = [](const double* a, const double* b) { double s = 0; for (int i = 0; i < 4; i++) s += a[i] * a[i]; for (int i = 0; i < 3; i++) s += b[i] * b[i]; return std::sqrt(s); };
  double x_norm = norm7(xq, xt);
  // gradient (unscaled) and Jacobi scaling from the first Jacobian
  auto gradient_max = [&](const std::vector<double>& rr, const std::vector<double>& JJ, const double* scale) {
    double gmax = 0;
    for (int j = 0; j < 6; j++) {
      double g = 0;
      for (int i = 0; i < m; i++) g += JJ[(size_t)i * 6 + j] * rr[i];
      if (scale) g /= scale[j];
      gmax = std::max(gmax, std::fabs(g));
    }
    return gmax;
  };
  double scale[6];
  for (int j = 0; j < 6; j++) {
    double s = 0;
    for (int i = 0; i < m; i++) s += J[(size_t)i * 6 + j] * J[(size_t)i * 6 + j];
    scale[j] = 1.0 / (1.0 + std::sqrt(s));
  }
  if (gradient_max(r, J, nullptr) <= 1e-10) { T.termination = 3; return; }
  for (int i = 0; i < m; i++) for (int j = 0; j < 6; j++) J[(size_t)i * 6 + j] *= scale[j];
  double radius = 1e4, decrease_factor = 2.0;
  bool reuse_diagonal = false;
  double diag[6];
  int invalid_run = 0;
  int iter = 0;
  T.termination = 0;
  while (true) {
    if (iter >= 4) { T.termination = 0; break; }                 // max_num_iterations (:214)
    if (radius < 1e-32) { T.termination = 6; break; }
    iter++;
    T.iterations = iter;
    T.radius[iter] = radius;
    if (!reuse_diagonal) {
      for (int j = 0; j < 6; j++) {
        double s = 0;
        for (int i = 0; i < m; i++) s += J[(size_t)i * 6 + j] * J[(size_t)i * 6 + j];
        diag[j] = std::min(std::max(s, 1e-6), 1e32);
      }
    }
    std::vector<double> A((size_t)(m + 6) * 6, 0.0), bb(m + 6, 0.0);
    std::copy(J.begin(), J.end(), A.begin());
    std::copy(r.begin(), r.end(), bb.begin());
    for (int j = 0; j < 6; j++) A[(size_t)(m + j) * 6 + j] = std::sqrt(diag[j] / radius);
    double y[6];
    bool ok = qr_solve6(A, bb, m + 6, y);
    reuse_diagonal = true;
    double step[6];
    double model_cost_change = 0;
    if (ok) {
      for (int j = 0; j < 6; j++) step[j] = -y[j];
      double acc = 0;
      for (int i = 0; i < m; i++) {
        double mr = 0;
        for (int j = 0; j < 6; j++) mr += J[(size_t)i * 6 + j] * step[j];
        acc += mr * (r[i] + mr / 2.0);
      }
      model_cost_change = -acc;
    }
    if (!ok || !(model_cost_change > 0.0)) {
      // TrustRegionMinimizer::HandleInvalidStep: max_num_consecutive_invalid_steps = 5, then
      // LevenbergMarquardtStrategy::StepIsInvalid(): radius *= 0.5, reuse_diagonal = true
      // (decrease_factor is only touched by StepRejected / StepAccepted)
      if (++invalid_run >= 5) { T.termination = 7; break; }
      radius = radius * 0.5; reuse_diagonal = true;
      continue;
    }
    invalid_run = 0;
    double delta[6];
    for (int j = 0; j < 6; j++) delta[j] = step[j] * scale[j];
    double cq[4], ct[3];
    quat_plus(xq, delta, cq);
    for (int k = 0; k < 3; k++) ct[k] = xt[k] + delta[3 + k];
    double cand_cost;
    std::vector<double> dummy_r, dummy_J;
    if (!evaluate(cq, ct, false, &cand_cost, dummy_r, dummy_J)) cand_cost = std::numeric_limits<double>::max();
    T.cost[iter] = cand_cost;
    double dq[4] = {xq[0] - cq[0], xq[1] - cq[1], xq[2] - cq[2], xq[3] - cq[3]};
    double dt[3] = {xt[0] - ct[0], xt[1] - ct[1], xt[2] - ct[2]};
    double step_norm = norm7(dq, dt);
    if (step_norm <= 1e-8 * (x_norm + 1e-8)) { T.termination = 1; break; }
    double cost_change = x_cost - cand_cost;
    if (std::fabs(cost_change) <= 1e-6 * x_cost) {
      T.termination = 2;
      if (apply_on_ftol && cost_change > 0) {
        for (int k = 0; k < 4; k++) xq[k] = cq[k];
        for (int k = 0; k < 3; k++) xt[k] = ct[k];
        x_cost = cand_cost;
      }
      break;
    }
    double rel = cost_change / model_cost_change;
    if (rel > 1e-3) {
      for (int k = 0; k < 4; k++) xq[k] = cq[k];
      for (int k = 0; k < 3; k++) xt[k] = ct[k];
      x_norm = norm7(xq, xt);
      if (!evaluate(xq, xt, true, &x_cost, r, J)) { T.termination = 5; break; }
      double gm = gradient_max(r, J, nullptr);
      for (int i = 0; i < m; i++) for (int j = 0; j < 6; j++) J[(size_t)i * 6 + j] *= scale[j];
      T.step_ok[iter] = 1; T.accepted++;
      double f = 1.0 - std::pow(2.0 * rel - 1.0, 3);
      radius = radius / std::max(1.0 / 3.0, f);
      radius = std::min(1e16, radius);
      decrease_factor = 2.0; reuse_diagonal = false;
      if (gm <= 1e-10) { T.termination = 3; break; }
    } else {
      radius = radius / decrease_factor; decrease_factor *= 2.0; reuse_diagonal = true;
    }
  }
  T.final_cost = x_cost;
  for (int k = 0; k < 4; k++) q[k] = xq[k];
  for (int k = 0; k < 3; k++) t[k] = xt[k];
}

// ------------------------------------------------------------------------------------------
// PCL 1.10 VoxelGrid::applyFilter for PointXYZI with all fields down-sampled
// (src/laser_odometry.cc:288-292, src/map.cc:56-60).  One centroid per occupied
// leaf, output ordered by ascending leaf id; float accumulation as Eigen::VectorXf centroid.
// ------------------------------------------------------------------------------------------
void voxel_grid(const std::vector<P4>& in, float leaf, std::vector<P4>& out) {
  out.clear();
  if (in.empty()) return;
  const float inv = 1.0f / leaf;
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (const P4& p : in) {
    if (!std::isfinite(p.x) || !std::isfinite(p.y) || !std::isfinite(p.z)) continue;
    mn[0] = std::min(mn[0], p.x); mn[1] = std::min(mn[1], p.y); mn[2] = std::min(mn[2], p.z);
    mx[0] = std::max(mx[0], p.x); mx[1] = std::max(mx[1], p.y); mx[2] = std::max(mx[2], p.z);
  }
  int minb[3], maxb[3], divb[3];
  for (int d = 0; d < 3; d++) {
    minb[d] = (int)std::floor(mn[d] * inv);
    maxb[d] = (int)std::floor(mx[d] * inv);
    divb[d] = maxb[d] - minb[d] + 1;
  }
  const int mul[3] = {1, divb[0], divb[0] * divb[1]};
  struct IdxPt { unsigned idx; unsigned pt; };
  std::vector<IdxPt> ip;
  ip.reserve(in.size());
  for (size_t i = 0; i < in.size(); i++) {
    const P4& p = in[i];
    if (!std::isfinite(p.x) || !std::isfinite(p.y) || !std::isfinite(p.z)) continue;
    int i0 = (int)std::floor(p.x * inv) - minb[0];
    int i1 = (int)std::floor(p.y * inv) - minb[1];
    int i2 = (int)std::floor(p.z * inv) - minb[2];
    ip.push_back({(unsigned)(i0 * mul[0] + i1 * mul[1] + i2 * mul[2]), (unsigned)i});
  }
  std::stable_sort(ip.begin(), ip.end(), [](const IdxPt& a, const IdxPt& b) { return a.idx < b.idx; });
  size_t k = 0;
  while (k < ip.size()) {
    size_t e = k;
    float sx = 0, sy = 0, sz = 0, si = 0;
    while (e < ip.size() && ip[e].idx == ip[k].idx) {
      const P4& p = in[ip[e].pt];
      sx += p.x; sy += p.y; sz += p.z; si += p.i;
      e++;
    }
    const float cnt = (float)(e - k);
    out.push_back({sx / cnt, sy / cnt, sz / cnt, si / cnt});
    k = e;
  }
}

// ------------------------------------------------------------------------------------------
// A12-A14  Map / Cell / HashKey                  src/map.cc:24-189, include/liodom/map.h:58-116
// Coarse hash cells (default 40 x 40 x 50 m) each holding a VoxelGrid(resolution)-filtered cloud.
// The unordered_map only serves lookups; iteration uses cells_vector_ (insertion order), so an
// ordered map is an equivalent restatement.
// ------------------------------------------------------------------------------------------
struct MapCell { std::vector<P4> points; bool modified = false; };
struct MapO {
  double voxel_xysize, inv_voxel_xysize, voxel_xysize_half, voxel_zsize, inv_voxel_zsize, voxel_zsize_half;
  float resolution;
  std::map<std::tuple<int, int, int>, int> cells;     // HashMap cells_ (map.h:91)
  std::vector<MapCell> cells_vector;                  // cells_vector_ (map.h:115)
  MapO(double xy, double z, double res)
      : voxel_xysize(xy), inv_voxel_xysize(1.0 / xy), voxel_xysize_half(xy / 2.0), voxel_zsize(z),
        inv_voxel_zsize(1.0 / z), voxel_zsize_half(z / 2.0), resolution((float)res) {}    // map.cc:70-81

  void updateMap(const std::vector<P4>& pc_in, const Iso& pose) {                     // map.cc:90-129
    for (size_t i = 0; i < pc_in.size(); i++) {
      const P4 point = transform_point(pose, pc_in[i]);                               // :93-94
      int voxel_x = int(std::floor(point.x * inv_voxel_xysize) * voxel_xysize + (voxel_xysize_half));   // :103
      int voxel_y = int(std::floor(point.y * inv_voxel_xysize) * voxel_xysize + (voxel_xysize_half));   // :104
      int voxel_z = int(std::floor(point.z * inv_voxel_zsize) * voxel_zsize + (voxel_zsize_half));      // :105
      auto key = std::make_tuple(voxel_x, voxel_y, voxel_z);
      auto it = cells.find(key);
      int ci;
      if (it == cells.end()) {                                                        // :110-114
        ci = (int)cells_vector.size();
        cells[key] = ci;
        cells_vector.push_back(MapCell());
      } else {
        ci = it->second;                                                              // :117
      }
      cells_vector[ci].points.push_back(point);                                       // :120
      cells_vector[ci].modified = true;
    }
    for (size_t i = 0; i < cells_vector.size(); i++) {                                // :124-128
      if (cells_vector[i].modified) {
        std::vector<P4> f;
        voxel_grid(cells_vector[i].points, resolution, f);                            // Cell::filter, :56-60
        cells_vector[i].points.swap(f);
        cells_vector[i].modified = false;
      }
    }
  }

  void getLocalMap(const Iso& pose, int cells_xy, int cells_z, std::vector<P4>& total_points) const {   // map.cc:141-189
    total_points.clear();
    int x = (int)pose.m[3];                                                           // :144 (truncation)
    int voxel_x = int(std::floor(x * inv_voxel_xysize) * voxel_xysize + (voxel_xysize_half));
    int y = (int)pose.m[7];                                                           // :147
    int voxel_y = int(std::floor(y * inv_voxel_xysize) * voxel_xysize + (voxel_xysize_half));
    int z = (int)pose.m[11];                                                          // :150
    int voxel_z = int(std::floor(z * inv_voxel_zsize) * voxel_zsize + (voxel_zsize_half));
    int init_x = voxel_x - cells_xy * voxel_xysize;                                   // :157-160
    int end_x = voxel_x + cells_xy * voxel_xysize;
    int init_y = voxel_y - cells_xy * voxel_xysize;
    int end_y = voxel_y + cells_xy * voxel_xysize;
    for (int i = init_x; i <= end_x; i += voxel_xysize) {                             // :162
      for (int j = init_y; j <= end_y; j += voxel_xysize) {                           // :163
        auto it = cells.find(std::make_tuple(i, j, voxel_z));
        if (it != cells.end()) {
          const std::vector<P4>& c = cells_vector[it->second].points;
          total_points.insert(total_points.end(), c.begin(), c.end());                // :169
        }
      }
    }
    int init_z = voxel_z - cells_z * voxel_xysize;                                    // :175 (xy size: reference quirk)
    int end_z = voxel_z + cells_z * voxel_xysize;                                     // :176
    for (int i = init_z; i <= end_z; i += voxel_zsize) {                              // :178
      auto it = cells.find(std::make_tuple(voxel_x, voxel_y, i));
      if (it != cells.end()) {
        const std::vector<P4>& c = cells_vector[it->second].points;
        total_points.insert(total_points.end(), c.begin(), c.end());                  // :184
      }
    }
  }
  size_t total() const { size_t n = 0; for (const auto& c : cells_vector) n += c.points.size(); return n; }
};

// ------------------------------------------------------------------------------------------
// A6  LocalMapManager                            src/laser_odometry.cc:24-69
// ------------------------------------------------------------------------------------------
struct LocalMapManager {
  std::vector<P4> total_points;
  size_t nframes = 0, max_nframes = 5;
  std::queue<size_t> sizes;
  void addPointCloud(const std::vector<P4>& pc) {
    total_points.insert(total_points.end(), pc.begin(), pc.end());                     // :36
    nframes++;                                                                         // :37
    sizes.push(pc.size());                                                             // :38
    if (nframes > max_nframes) {                                                       // :41
      size_t pc_size = sizes.front();                                                  // :43
      sizes.pop();
      total_points.erase(total_points.begin(), total_points.begin() + pc_size);         // :47-55
      nframes--;                                                                       // :58
    }
  }
};

struct StepInfo {
  int32_t n_edges, map_points, matches[2];
  LmTrace lm[2];
};

// ------------------------------------------------------------------------------------------
// A7-A9  LaserOdometer (steady-state branch)     src/laser_odometry.cc:100-366
// ------------------------------------------------------------------------------------------
struct Odometer {
  orc_params_t prm;
  bool init = false;
  Iso prev_odom = iso_identity(), odom = iso_identity();
  double param_q[4] = {0, 0, 0, 1}, param_t[3] = {0, 0, 0};
  LocalMapManager lmap;
  std::vector<P4> received_map;                       // SharedData::setLocalMap (mapClb)
  bool use_imu = false;                               // params->use_imu_ (params.cc:96)
  double imu_q[4] = {0, 0, 0, 1};                     // SharedData::last_IMU_ori_ (imuClb, liodom_node.cc:66-70), [x y z w]
  Iso laser_to_base = iso_identity();                 // laser_to_base_ (laser_odometry.cc:110-119)
  // mapping = true: the liodom_mapping node (src/liodom_mapping_node.cc:45-90, defaults :115-134)
  // replayed synchronously — the reference is asynchronous / non-deterministic here (SURVEY.md
  // §3.3): after scan k, updateMap(edges_k, pose_k); scan k+1 receives getLocalMap(pose_k, 2, 1).
  MapO mapper{40.0, 50.0, 0.4};
  int cells_xy = 2, cells_z = 1;
  // debug capture of the last step
  std::vector<int32_t> corr_valid[2], corr_a[2], corr_b[2];
  std::vector<P4> queries[2];                         // world-frame float queries of the last step (:307-308)
  StepInfo info{};

  explicit Odometer(const orc_params_t& p) : prm(p) { lmap.max_nframes = (size_t)p.local_map_size; }

  // One edge of addEdgeConstraints' loop (:323-357): 5-NN, distance gate, line gate; the line points are NN0 / NN1.
  bool match_edge(const std::vector<P4>& local_map, KdTree& tree, const P4& q, int* ia, int* ib) const {
    Knn5 nn;
    if (prm.knn_mode == 1) tree.search(q, nn); else knn5_brute(local_map, q, nn);      // :323
    if (nn.n < 5) return false;                  // reference would read sq_dist[4] out of bounds (UB)
    if (!(nn.d[4] < 1.0)) return false;                                                // :324
    double center[3] = {0, 0, 0};
    double nc[5][3];
    for (int j = 0; j < 5; j++) {                                                      // :327-333
      nc[j][0] = local_map[nn.idx[j]].x; nc[j][1] = local_map[nn.idx[j]].y; nc[j][2] = local_map[nn.idx[j]].z;
      center[0] = center[0] + nc[j][0]; center[1] = center[1] + nc[j][1]; center[2] = center[2] + nc[j][2];
    }
    center[0] = center[0] / 5.0; center[1] = center[1] / 5.0; center[2] = center[2] / 5.0;  // :334
    double cov[6] = {0, 0, 0, 0, 0, 0};                                                // :336
    for (int j = 0; j < 5; j++) {                                                      // :337-340
      double zx = nc[j][0] - center[0], zy = nc[j][1] - center[1], zz = nc[j][2] - center[2];
      cov[0] = cov[0] + zx * zx; cov[1] = cov[1] + zx * zy; cov[2] = cov[2] + zx * zz;
      cov[3] = cov[3] + zy * zy; cov[4] = cov[4] + zy * zz; cov[5] = cov[5] + zz * zz;
    }
    double ev[3];
    eig3_sym(cov, ev);                                                                 // :342
    if (!(ev[2] > 3 * ev[1])) return false;                                            // :344
    *ia = nn.idx[0]; *ib = nn.idx[1];                                                  // :351-357
    return true;
  }

  // A9 addEdgeConstraints (:300-366)
  void add_edge_constraints(const std::vector<P4>& edges, const std::vector<P4>& local_map_gen,
                            const Iso& pose, std::vector<Corr>& blocks, int it) {
    std::vector<P4> edges_map(edges.size());
    for (size_t i = 0; i < edges.size(); i++) edges_map[i] = transform_point(pose, edges[i]);  // :307-308
    std::vector<P4> local_map(local_map_gen);                                          // :310-311
    if (prm.mapping) local_map.insert(local_map.end(), received_map.begin(), received_map.end());  // :312-314
    KdTree tree;
    if (prm.knn_mode == 1) tree.build(local_map);                                      // :318-319
    corr_valid[it].assign(edges.size(), 0);
    corr_a[it].assign(edges.size(), -1);
    corr_b[it].assign(edges.size(), -1);
    int correct = 0;
    queries[it] = edges_map;
    for (size_t i = 0; i < edges_map.size(); i++) {                                    // :320
      int ia = -1, ib = -1;
      if (match_edge(local_map, tree, edges_map[i], &ia, &ib)) {
        correct++;
        Corr c;
        c.p[0] = edges[i].x; c.p[1] = edges[i].y; c.p[2] = edges[i].z;               // :347-349
        c.a[0] = local_map[ia].x; c.a[1] = local_map[ia].y; c.a[2] = local_map[ia].z;  // :351-353
        c.b[0] = local_map[ib].x; c.b[1] = local_map[ib].y; c.b[2] = local_map[ib].z;  // :355-357
        blocks.push_back(c);                                                         // :359-360
        corr_valid[it][i] = 1; corr_a[it][i] = ia; corr_b[it][i] = ib;
      }
    }
    info.matches[it] = correct;
  }

  // One iteration of LaserOdometer::operator() (:107-267) for one edge cloud.
  void step(const std::vector<P4>& feats, double pose_out[7]) {
    info = StepInfo{};
    info.n_edges = (int)feats.size();
    if (!init) {                                                                       // :108
      lmap.addPointCloud(feats);                                                       // :123
      init = true;                                                                     // :124
      info.map_points = (int)lmap.total_points.size();
    } else {
      // computeLocalMap (:274-298)
      std::vector<P4> gen;
      if (prm.filter_local_map && lmap.nframes == (size_t)prm.local_map_size && !prm.mapping) {  // :286
        voxel_grid(lmap.total_points, 0.4f, gen);                                      // :288-292
      } else {
        gen = lmap.total_points;                                                       // :294
      }
      info.map_points = (int)gen.size() + (prm.mapping ? (int)received_map.size() : 0);
      // predict (:148-150)
      Iso pred = iso_mul(odom, iso_mul(iso_inverse(prev_odom), odom));
      prev_odom = odom;
      odom = pred;
      if (use_imu) odom = imu_override(odom, imu_q, laser_to_base, prm.pose_rotation_mode);   // :152-183
      // initial guess (:186-195)
      quat_from_rot(rotation_of(odom, prm.pose_rotation_mode), param_q);               // :186 q_curr(odom_.rotation())
      param_t[0] = odom.m[3]; param_t[1] = odom.m[7]; param_t[2] = odom.m[11];
      for (int optim_it = 0; optim_it < 2; optim_it++) {                               // :198
        std::vector<Corr> blocks;
        add_edge_constraints(feats, gen, odom, blocks, optim_it);                      // :209
        lm_solve(blocks, param_q, param_t, prm.min_range, prm.max_range,
                 prm.lm_apply_step_on_ftol, &info.lm[optim_it]);                       // :212-218
        odom = iso_from_qt(param_q, param_t);                                          // :222-227
      }
      std::vector<P4> edges_map(feats.size());
      for (size_t i = 0; i < feats.size(); i++) edges_map[i] = transform_point(odom, feats[i]);  // :231-232
      lmap.addPointCloud(edges_map);                                                   // :235
    }
    if (prm.mapping == 1) {     // mapping == 2: the ~map cloud only comes from orc_odom_set_received_map (external mapper)
      mapper.updateMap(feats, odom);                          // liodom_mapping_node.cc:69
      mapper.getLocalMap(odom, cells_xy, cells_z, received_map);   // :82 -> mapClb (liodom_node.cc:57-64)
    }
    // pose as published: Quaterniond((odom_ * laser_to_base_).rotation()) (publishOdom :402-403);
    // laser_to_base_ is left out here as in the C-ABI (poses are returned in the laser frame)
    double q[4];
    quat_from_rot(rotation_of(odom, prm.pose_rotation_mode), q);
    pose_out[0] = q[0]; pose_out[1] = q[1]; pose_out[2] = q[2]; pose_out[3] = q[3];
    pose_out[4] = odom.m[3]; pose_out[5] = odom.m[7]; pose_out[6] = odom.m[11];
  }
};

}  // namespace

// ==========================================================================================
// C interface (ctypes)
// ==========================================================================================
extern "C" {

// Ring split: order[] receives the source indices sorted by ring (stable), ring_offsets[H+1].
int orc_split(const orc_params_t* p, const float* xyzi, int64_t n, int height, int width,
              int32_t* ring_offsets, int32_t* order) {
  std::vector<std::vector<int32_t>> rings;
  split_point_cloud(*p, reinterpret_cast<const P4*>(xyzi), n, height, width, rings);
  int32_t off = 0;
  for (int r = 0; r < p->scan_lines; r++) {
    ring_offsets[r] = off;
    for (int32_t s : rings[r]) order[off++] = s;
  }
  ring_offsets[p->scan_lines] = off;
  return off;
}

// Full extraction (A1-A5).  Returns the number of edges (or -needed if cap is too small).
// curv (optional, n doubles): smoothness per compacted point in ring-sorted order (NaN where
// undefined or where the ring was skipped).
int orc_extract(const orc_params_t* p, const float* xyzi, int64_t n, int height, int width,
                float* edges_xyzi, int32_t* edge_ring, int32_t* edge_idx_in_ring,
                int32_t* edge_src, int cap, double* curv) {
  const P4* pc = reinterpret_cast<const P4*>(xyzi);
  std::vector<std::vector<int32_t>> rings;
  split_point_cloud(*p, pc, n, height, width, rings);
  EdgeOut out;
  std::vector<std::vector<double>> cv;
  extract_features(*p, pc, rings, out, curv ? &cv : nullptr);
  if (curv) {
    int64_t off = 0;
    for (int r = 0; r < p->scan_lines; r++) {
      for (size_t k = 0; k < rings[r].size(); k++)
        curv[off + k] = k < cv[r].size() ? cv[r][k] : std::numeric_limits<double>::quiet_NaN();
      off += (int64_t)rings[r].size();
    }
  }
  const int ne = (int)out.pts.size();
  if (ne > cap) return -ne;
  for (int i = 0; i < ne; i++) {
    std::memcpy(edges_xyzi + 4 * i, &out.pts[i], sizeof(P4));
    if (edge_ring) edge_ring[i] = out.ring[i];
    if (edge_idx_in_ring) edge_idx_in_ring[i] = out.idx_in_ring[i];
    if (edge_src) edge_src[i] = out.src[i];
  }
  return ne;
}

void* orc_odom_create(const orc_params_t* p) { return new Odometer(*p); }
void orc_odom_destroy(void* h) { delete static_cast<Odometer*>(h); }

struct orc_lm_trace_t {
  int32_t iterations, accepted, termination, pad;
  double initial_cost, final_cost;
  double cost[5], radius[5];
  int32_t step_ok[5]; int32_t pad2;
};
struct orc_step_info_t {
  int32_t n_edges, map_points, matches[2];
  orc_lm_trace_t lm[2];
};

static void copy_trace(const LmTrace& s, orc_lm_trace_t* d) {
  d->iterations = s.iterations; d->accepted = s.accepted; d->termination = s.termination; d->pad = 0;
  d->initial_cost = s.initial_cost; d->final_cost = s.final_cost;
  for (int i = 0; i < 5; i++) { d->cost[i] = s.cost[i]; d->radius[i] = s.radius[i]; d->step_ok[i] = s.step_ok[i]; }
  d->pad2 = 0;
}

// One odometry step on an edge cloud (sensor frame).  pose_out = [qx qy qz qw tx ty tz].
int orc_odom_step(void* h, const float* edges_xyzi, int n_edges, double* pose_out,
                  orc_step_info_t* info) {
  Odometer* o = static_cast<Odometer*>(h);
  std::vector<P4> feats(n_edges);
  if (n_edges) std::memcpy(feats.data(), edges_xyzi, sizeof(P4) * (size_t)n_edges);
  o->step(feats, pose_out);
  if (info) {
    info->n_edges = o->info.n_edges; info->map_points = o->info.map_points;
    info->matches[0] = o->info.matches[0]; info->matches[1] = o->info.matches[1];
    copy_trace(o->info.lm[0], &info->lm[0]); copy_trace(o->info.lm[1], &info->lm[1]);
  }
  return 0;
}

// Correspondences of the last step, outer iteration `it`: valid flag and map indices of the two
// line points per edge.
int orc_odom_last_corr(void* h, int it, int32_t* valid, int32_t* ia, int32_t* ib, int cap) {
  Odometer* o = static_cast<Odometer*>(h);
  int n = (int)o->corr_valid[it].size();
  if (n > cap) return -n;
  for (int i = 0; i < n; i++) { valid[i] = o->corr_valid[it][i]; ia[i] = o->corr_a[it][i]; ib[i] = o->corr_b[it][i]; }
  return n;
}

// World-frame float queries of the last step, outer iteration `it` (xyzi, n_edges rows).
int orc_odom_last_queries(void* h, int it, float* xyzi, int cap) {
  Odometer* o = static_cast<Odometer*>(h);
  const int n = (int)o->queries[it].size();
  if (n > cap) return -n;
  if (n) std::memcpy(xyzi, o->queries[it].data(), sizeof(P4) * (size_t)n);
  return n;
}
// addEdgeConstraints' per-edge loop (:320-361) on explicit inputs: a local-map cloud and world-frame float
// queries.  valid / ia / ib per query.  mode 0 brute force, 1 kd-tree.
void orc_match_edges(const orc_params_t* p, const float* map_xyzi, int64_t m, const float* q_xyzi, int64_t nq,
                     int32_t* valid, int32_t* ia, int32_t* ib) {
  Odometer od(*p);
  std::vector<P4> map((size_t)m);
  if (m) std::memcpy(map.data(), map_xyzi, sizeof(P4) * (size_t)m);
  KdTree tree;
  if (p->knn_mode == 1) tree.build(map);
  const P4* q = reinterpret_cast<const P4*>(q_xyzi);
  for (int64_t i = 0; i < nq; i++) {
    int a = -1, b = -1;
    const bool ok = od.match_edge(map, tree, q[i], &a, &b);
    valid[i] = ok ? 1 : 0; ia[i] = ok ? a : -1; ib[i] = ok ? b : -1;
  }
}
int64_t orc_odom_window_size(void* h) { return (int64_t)static_cast<Odometer*>(h)->lmap.total_points.size(); }
int orc_odom_window_frames(void* h) { return (int)static_cast<Odometer*>(h)->lmap.nframes; }
int64_t orc_odom_get_window(void* h, float* xyzi, int64_t cap) {
  Odometer* o = static_cast<Odometer*>(h);
  int64_t n = (int64_t)o->lmap.total_points.size();
  if (n > cap) return -n;
  if (n) std::memcpy(xyzi, o->lmap.total_points.data(), sizeof(P4) * (size_t)n);
  return n;
}
void orc_odom_set_imu(void* h, int use_imu, const double* q_xyzw) {
  Odometer* o = static_cast<Odometer*>(h);
  o->use_imu = use_imu != 0;
  if (q_xyzw) for (int k = 0; k < 4; k++) o->imu_q[k] = q_xyzw[k];
}
void orc_odom_set_laser_to_base(void* h, const double* T12) {
  Odometer* o = static_cast<Odometer*>(h);
  for (int k = 0; k < 12; k++) o->laser_to_base.m[k] = T12[k];
}
// the override alone (unit tests): T12 in, T12 out
void orc_imu_override(const double* T12, const double* imu_q, const double* l2b12, double* out12, int rotation_mode) {
  Iso T, L;
  for (int k = 0; k < 12; k++) { T.m[k] = T12[k]; L.m[k] = l2b12[k]; }
  const Iso R = imu_override(T, imu_q, L, rotation_mode);
  for (int k = 0; k < 12; k++) out12[k] = R.m[k];
}
// LaserOdometer::publishOdom (src/laser_odometry.cc:395-436): the nav_msgs/Odometry numbers.
// out[0..3] orientation x y z w, out[4..6] position, out[7..9] twist.linear, out[10..12] twist.angular
void orc_publish_odom(const double* prev12, const double* cur12, const double* l2b12, double delta_time, double* out, int rotation_mode) {
  Iso P, T, L;
  for (int k = 0; k < 12; k++) { P.m[k] = prev12[k]; T.m[k] = cur12[k]; L.m[k] = l2b12[k]; }
  const Iso odom_base_link = iso_mul(T, L);                                   // :403
  quat_from_rot(rotation_of(odom_base_link, rotation_mode), out);             // :403 q_current(odom_base_link.rotation())
  out[4] = odom_base_link.m[3]; out[5] = odom_base_link.m[7]; out[6] = odom_base_link.m[11];   // :405
  const Iso delta_odom = iso_mul(iso_inverse(iso_mul(P, L)), odom_base_link); // :416
  out[7] = delta_odom.m[3] / delta_time; out[8] = delta_odom.m[7] / delta_time; out[9] = delta_odom.m[11] / delta_time;  // :417-420
  double qd[4], roll, pitch, yaw;
  quat_from_rot(rotation_of(delta_odom, rotation_mode), qd);                  // :420 q_delta(delta_odom.rotation())
  tf_get_rpy(tf_matrix_from_quat(qd), &roll, &pitch, &yaw);                   // :423-426
  out[10] = roll / delta_time; out[11] = pitch / delta_time; out[12] = yaw / delta_time;       // :427-429
}
void orc_odom_set_received_map(void* h, const float* xyzi, int64_t n) {
  Odometer* o = static_cast<Odometer*>(h);
  o->received_map.resize((size_t)n);
  if (n) std::memcpy(o->received_map.data(), xyzi, sizeof(P4) * (size_t)n);
}
int64_t orc_odom_get_received_map(void* h, float* xyzi, int64_t cap) {
  Odometer* o = static_cast<Odometer*>(h);
  int64_t n = (int64_t)o->received_map.size();
  if (n > cap) return -n;
  if (n) std::memcpy(xyzi, o->received_map.data(), sizeof(P4) * (size_t)n);
  return n;
}
// --- Map (mapping node) unit-level entry points ---------------------------------------------
void* orc_map_create(double xy, double z, double res) { return new MapO(xy, z, res); }
void orc_map_destroy(void* m) { delete static_cast<MapO*>(m); }
void orc_map_update(void* m, const float* xyzi, int64_t n, const double* T12) {
  std::vector<P4> pc((size_t)n);
  if (n) std::memcpy(pc.data(), xyzi, sizeof(P4) * (size_t)n);
  Iso T; std::memcpy(T.m, T12, sizeof(double) * 12);
  static_cast<MapO*>(m)->updateMap(pc, T);
}
int64_t orc_map_get_local(void* m, const double* T12, int cells_xy, int cells_z, float* out, int64_t cap) {
  Iso T; std::memcpy(T.m, T12, sizeof(double) * 12);
  std::vector<P4> o;
  static_cast<MapO*>(m)->getLocalMap(T, cells_xy, cells_z, o);
  if ((int64_t)o.size() > cap) return -(int64_t)o.size();
  if (!o.empty()) std::memcpy(out, o.data(), sizeof(P4) * o.size());
  return (int64_t)o.size();
}
int64_t orc_map_get_all(void* m, float* out, int64_t cap) {            // Map::getMap, map.cc:131-139
  MapO* M = static_cast<MapO*>(m);
  int64_t n = 0;
  for (const auto& c : M->cells_vector) {
    if (n + (int64_t)c.points.size() > cap) return -1;
    if (!c.points.empty()) std::memcpy(out + 4 * n, c.points.data(), sizeof(P4) * c.points.size());
    n += (int64_t)c.points.size();
  }
  return n;
}
int orc_map_num_cells(void* m) { return (int)static_cast<MapO*>(m)->cells_vector.size(); }
int64_t orc_odom_map_total(void* h) { return (int64_t)static_cast<Odometer*>(h)->mapper.total(); }
void orc_odom_get_state(void* h, double* odom12, double* prev12) {
  Odometer* o = static_cast<Odometer*>(h);
  std::memcpy(odom12, o->odom.m, sizeof(double) * 12);
  std::memcpy(prev12, o->prev_odom.m, sizeof(double) * 12);
}

// --- unit-level entry points ---------------------------------------------------------------
void orc_knn5(const float* map_xyzi, int64_t m, const float* q_xyzi, int64_t nq, int mode,
              int32_t* idx, float* dist) {
  std::vector<P4> map((size_t)m);
  if (m) std::memcpy(map.data(), map_xyzi, sizeof(P4) * (size_t)m);
  KdTree tree;
  if (mode == 1) tree.build(map);
  const P4* q = reinterpret_cast<const P4*>(q_xyzi);
  for (int64_t i = 0; i < nq; i++) {
    Knn5 k;
    if (mode == 1) tree.search(q[i], k); else knn5_brute(map, q[i], k);
    for (int j = 0; j < 5; j++) { idx[5 * i + j] = j < k.n ? k.idx[j] : -1; dist[5 * i + j] = j < k.n ? k.d[j] : INFINITY; }
  }
}
void orc_eig3(const double* a6, double* ev3) { eig3_sym(a6, ev3); }
// residual r[3] and local Jacobian J[18] (3x6 row-major, q tangent first) WITHOUT the loss
// correction (raw), plus the Huber-corrected cost contribution.
int orc_point2line(const double* q, const double* t, const double* p, const double* a,
                   const double* b, double min_d, double max_d, double* r, double* J,
                   double* Jglobal21) {
  Corr c; for (int i = 0; i < 3; i++) { c.p[i] = p[i]; c.a[i] = a[i]; c.b[i] = b[i]; }
  Jet res[3];
  point2line_jets(c, q, t, min_d, max_d, res);
  double P[12];
  quat_plus_jacobian(q, P);
  for (int i = 0; i < 3; i++) {
    r[i] = res[i].a;
    for (int k = 0; k < 3; k++) {
      double acc = 0;
      for (int m = 0; m < 4; m++) acc += res[i].v[m] * P[m * 3 + k];
      J[i * 6 + k] = acc;
    }
    for (int k = 0; k < 3; k++) J[i * 6 + 3 + k] = res[i].v[4 + k];
    if (Jglobal21) for (int k = 0; k < 7; k++) Jglobal21[i * 7 + k] = res[i].v[k];
  }
  return 0;
}
void orc_quat_plus(const double* x, const double* delta, double* out) { quat_plus(x, delta, out); }
// Full LM solve on explicit correspondences (n blocks, each p[3] a[3] b[3] as 9 doubles).
int orc_lm_solve(const double* blocks9, int n, double* q, double* t, double min_d, double max_d,
                 int apply_on_ftol, orc_lm_trace_t* trace) {
  std::vector<Corr> blocks((size_t)n);
  for (int i = 0; i < n; i++)
    for (int k = 0; k < 3; k++) { blocks[i].p[k] = blocks9[9 * i + k]; blocks[i].a[k] = blocks9[9 * i + 3 + k]; blocks[i].b[k] = blocks9[9 * i + 6 + k]; }
  LmTrace tr;
  lm_solve(blocks, q, t, min_d, max_d, apply_on_ftol, &tr);
  if (trace) copy_trace(tr, trace);
  return tr.termination;
}
// Huber-corrected total cost 0.5*sum rho(|r|^2) at (q, t) — for the SciPy cross-check.
double orc_cost(const double* blocks9, int n, const double* q, const double* t, double min_d, double max_d) {
  double total = 0;
  for (int i = 0; i < n; i++) {
    Corr c; for (int k = 0; k < 3; k++) { c.p[k] = blocks9[9 * i + k]; c.a[k] = blocks9[9 * i + 3 + k]; c.b[k] = blocks9[9 * i + 6 + k]; }
    double c1, r[3], J[18];
    if (!eval_block(c, q, t, min_d, max_d, false, &c1, r, J)) return NAN;
    total += c1;
  }
  return total;
}
int64_t orc_voxel_grid(const float* xyzi, int64_t n, float leaf, float* out, int64_t cap) {
  std::vector<P4> in((size_t)n), o;
  if (n) std::memcpy(in.data(), xyzi, sizeof(P4) * (size_t)n);
  voxel_grid(in, leaf, o);
  if ((int64_t)o.size() > cap) return -(int64_t)o.size();
  if (!o.empty()) std::memcpy(out, o.data(), sizeof(P4) * o.size());
  return (int64_t)o.size();
}
void orc_transform(const double* T12, const float* in, int64_t n, float* out) {
  Iso T; std::memcpy(T.m, T12, sizeof(double) * 12);
  const P4* pi = reinterpret_cast<const P4*>(in);
  P4* po = reinterpret_cast<P4*>(out);
  for (int64_t i = 0; i < n; i++) po[i] = transform_point(T, pi[i]);
}
void orc_debug_fail_linear_solves(int n) { g_fail_linear_solves = n; }
// CPU-baseline thread policy (see g_stencil_threads): 0 keeps the current value.
void orc_set_threads(int stencil_threads, int eval_threads) {
  if (stencil_threads > 0) g_stencil_threads = stencil_threads;
  if (eval_threads > 0) g_eval_threads = eval_threads;
}
// Transform::rotation() alone (unit tests): T12 in -> T12 out
void orc_rotation_of(const double* T12, int mode, double* out12) {
  Iso T; for (int k = 0; k < 12; k++) T.m[k] = T12[k];
  const Iso R = rotation_of(T, mode);
  for (int k = 0; k < 12; k++) out12[k] = R.m[k];
}
void orc_pose_ops(const double* q_in, const double* t_in, double* T12, double* q_back) {
  Iso T = iso_from_qt(q_in, t_in);
  std::memcpy(T12, T.m, sizeof(double) * 12);
  quat_from_rot(T, q_back);
}

}  // extern "C"
