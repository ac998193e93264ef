// Synthetic LiDAR stream generator (bench / test input only; no reference code involved).
//
// Implements the synthetic input of SURVEY.md §8(d): a static analytic world (ground plane at
// z = -1.7 m plus axis-aligned boxes / thin pillars on a jittered grid), a smooth sensor
// trajectory (0.1 m/scan forward, constant yaw rate, small sinusoidal z / roll / pitch) and a
// per-scan ray cast of H x W rays with range-only Gaussian noise along the ray.  Rays without a
// hit give NaN points (exercises the reference's isValidPoint, feature_extractor.cc:84-102).
//
// Layouts handed out (packed float4 XYZI):
//   lidar_type 0 (Velodyne, unorganised): firing order, i = col*H + ring
//   lidar_type 1 (Ouster, organised):     row-major,    i = ring*W + col
// intensity = ring*10000 + col, so every extracted edge can be traced back to its source ray.
//
// Ring elevations are chosen to sit inside the elevation bins of the reference's ring split
// (feature_extractor.cc:130-148): HDL-64 upper block 1.95 - k/3 deg, lower block
// -8.78 - k/2 deg; VLP-16 -15 + 2k deg; HDL-32 (-92/3 + 4k/3 + 0.6) deg; 128 rows are
// row-indexed (lidar_type 1) and span +22.5 .. -22.5 deg.
//
// Yaw rate: 0.5 deg/scan, SURVEY.md §8(d) (110 deg over a 220-scan stream).  With Eigen 3.3's
// Transform::rotation() (polar factor, the default pose_rotation_mode) the reference's pose recursion
// is stable through any accumulated rotation; with the Eigen >= 3.4 alias semantics it loses track
// beyond ~90 deg (tests/test_oracle_odometry.py::test_rotation_mode_soak).
//
// Determinism: every random number is a pure function of (seed, counter); the output does not
// depend on the number of OpenMP threads.
#ifdef _OPENMP
#include <omp.h>
#endif
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

namespace {

struct Box { double lo[3], hi[3]; };

inline uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
inline double u01(uint64_t h) { return ((h >> 11) + 0.5) * (1.0 / 9007199254740992.0); }

struct World {
  std::vector<Box> boxes;
  uint32_t seed = 0;
  double radius = -1.0;
  bool built = false;
};

// Boxes on a jittered 10 m grid over [-160, 160]^2; nothing within 5.5 m of the sensor's
// trajectory (a circle of radius `radius` through the origin, or the x axis when radius <= 0) so
// the sensor never enters an obstacle.
void build_world(World& w, uint32_t seed, double radius) {
  w.boxes.clear();
  w.seed = seed;
  w.radius = radius;
  const double cell = 10.0;
  const int half = 16;
  const double traj_cx = 0.0, traj_cy = radius;  // circle centre
  for (int gy = -half; gy < half; ++gy) {
    for (int gx = -half; gx < half; ++gx) {
      uint64_t h = splitmix64(((uint64_t)seed << 32) ^ (uint64_t)((gy + 64) * 256 + (gx + 64)));
      double r0 = u01(h); h = splitmix64(h);
      if (r0 > 0.62) continue;
      double jx = u01(h); h = splitmix64(h);
      double jy = u01(h); h = splitmix64(h);
      double kind = u01(h); h = splitmix64(h);
      double s1 = u01(h); h = splitmix64(h);
      double s2 = u01(h); h = splitmix64(h);
      double s3 = u01(h); h = splitmix64(h);
      double cx = (gx + 0.2 + 0.6 * jx) * cell;
      double cy = (gy + 0.2 + 0.6 * jy) * cell;
      double dx = cx - traj_cx, dy = cy - traj_cy;
      const double clear = 5.5 + 3.0;   // corridor half-width + largest half-diagonal of a box
      if (radius > 0.0) { if (std::fabs(std::sqrt(dx * dx + dy * dy) - radius) < clear) continue; }
      else if (std::fabs(cy) < clear) continue;
      double wx, wy, hz;
      if (kind < 0.35) {            // thin pillar / pole
        wx = 0.15 + 0.25 * s1; wy = 0.15 + 0.25 * s2; hz = 3.0 + 9.0 * s3;
      } else if (kind < 0.8) {      // box / vehicle-sized
        wx = 0.8 + 2.5 * s1; wy = 0.8 + 2.5 * s2; hz = 1.0 + 3.0 * s3;
      } else {                      // building block
        wx = 3.0 + 3.5 * s1; wy = 3.0 + 3.5 * s2; hz = 5.0 + 12.0 * s3;
      }
      Box b;
      b.lo[0] = cx - 0.5 * wx; b.hi[0] = cx + 0.5 * wx;
      b.lo[1] = cy - 0.5 * wy; b.hi[1] = cy + 0.5 * wy;
      b.lo[2] = -1.7;          b.hi[2] = -1.7 + hz;
      w.boxes.push_back(b);
    }
  }
  w.built = true;
}

World g_world;

void ring_elevations(int height, std::vector<double>& el) {
  el.resize(height);
  if (height == 64) {
    for (int k = 0; k < 32; ++k) el[k] = 1.95 - k / 3.0;
    for (int k = 0; k < 32; ++k) el[32 + k] = -8.78 - k / 2.0;
  } else if (height == 32) {
    for (int k = 0; k < 32; ++k) el[k] = -92.0 / 3.0 + (k + 0.45) * 4.0 / 3.0;
  } else if (height == 16) {
    for (int k = 0; k < 16; ++k) el[k] = -15.0 + 2.0 * k;
  } else {
    for (int k = 0; k < height; ++k)
      el[k] = 22.5 - 45.0 * (k + 0.5) / height;
  }
}

struct Pose { double R[9]; double t[3]; };

// Ground-truth trajectory: world <- sensor at scan k of a stream.
void trajectory(int stream, int k, double yaw_rate_deg, double speed, Pose& P, double q_out[4]) {
  const double dyaw = yaw_rate_deg * M_PI / 180.0;
  // position integrates 0.1 m per scan along the current heading
  double x = 0, y = 0;
  for (int i = 0; i < k; ++i) { x += speed * std::cos(dyaw * i); y += speed * std::sin(dyaw * i); }
  double ph = 0.37 * stream;
  double z = 0.05 * std::sin(0.1 * k + ph);
  double yaw = dyaw * k;
  double roll = 0.01 * std::sin(0.07 * k + ph);
  double pitch = 0.008 * std::sin(0.05 * k + 1.0 + ph);
  double cy = std::cos(yaw), sy = std::sin(yaw);
  double cp = std::cos(pitch), sp = std::sin(pitch);
  double cr = std::cos(roll), sr = std::sin(roll);
  // R = Rz(yaw) * Ry(pitch) * Rx(roll)
  P.R[0] = cy * cp; P.R[1] = cy * sp * sr - sy * cr; P.R[2] = cy * sp * cr + sy * sr;
  P.R[3] = sy * cp; P.R[4] = sy * sp * sr + cy * cr; P.R[5] = sy * sp * cr - cy * sr;
  P.R[6] = -sp;     P.R[7] = cp * sr;                P.R[8] = cp * cr;
  P.t[0] = x; P.t[1] = y; P.t[2] = z;
  // quaternion (x, y, z, w) from yaw/pitch/roll
  double hy = 0.5 * yaw, hp = 0.5 * pitch, hr = 0.5 * roll;
  double c1 = std::cos(hy), s1 = std::sin(hy), c2 = std::cos(hp), s2 = std::sin(hp);
  double c3 = std::cos(hr), s3 = std::sin(hr);
  q_out[3] = c1 * c2 * c3 + s1 * s2 * s3;
  q_out[0] = c1 * c2 * s3 - s1 * s2 * c3;
  q_out[1] = c1 * s2 * c3 + s1 * c2 * s3;
  q_out[2] = s1 * c2 * c3 - c1 * s2 * s3;
}

inline double ray_box(const double o[3], const double d[3], const Box& b) {
  double tmin = 0.0, tmax = std::numeric_limits<double>::infinity();
  for (int a = 0; a < 3; ++a) {
    if (std::fabs(d[a]) < 1e-12) {
      if (o[a] < b.lo[a] || o[a] > b.hi[a]) return -1.0;
    } else {
      double inv = 1.0 / d[a];
      double t0 = (b.lo[a] - o[a]) * inv, t1 = (b.hi[a] - o[a]) * inv;
      if (t0 > t1) { double s = t0; t0 = t1; t1 = s; }
      if (t0 > tmin) tmin = t0;
      if (t1 < tmax) tmax = t1;
      if (tmin > tmax) return -1.0;
    }
  }
  return tmin > 0.0 ? tmin : -1.0;
}

}  // namespace

extern "C" {

struct synth_cfg_t {
  int32_t height;        // rings
  int32_t width;         // azimuth steps
  int32_t lidar_type;    // 0 firing order (Velodyne), 1 row-major organised (Ouster)
  uint32_t world_seed;   // static world layout
  double noise_sigma;    // range noise (m), along the ray
  double max_cast_range; // rays longer than this return NaN
  double yaw_rate_deg;   // yaw per scan (deg); default 0.5 (see header)
  double speed;          // forward motion per scan (m); SURVEY.md §8(d): 0.1
};

// Fills xyzi (height*width*4 floats) for scan `scan` of stream `stream`, and the ground-truth
// pose gt_pose = [qx qy qz qw tx ty tz] (world <- sensor).  Returns 0.
int synth_scan(const synth_cfg_t* cfg, int stream, int scan, float* xyzi, double* gt_pose) {
  if (!cfg || !xyzi || cfg->height <= 0 || cfg->width <= 0) return -1;
  const double radius = cfg->yaw_rate_deg != 0.0 ? cfg->speed / (cfg->yaw_rate_deg * M_PI / 180.0) : -1.0;
  if (!g_world.built || g_world.seed != cfg->world_seed || g_world.radius != radius) build_world(g_world, cfg->world_seed, radius);
  const int H = cfg->height, W = cfg->width;
  std::vector<double> el;
  ring_elevations(H, el);
  Pose P; double q[4];
  trajectory(stream, scan, cfg->yaw_rate_deg, cfg->speed, P, q);
  if (gt_pose) {
    gt_pose[0] = q[0]; gt_pose[1] = q[1]; gt_pose[2] = q[2]; gt_pose[3] = q[3];
    gt_pose[4] = P.t[0]; gt_pose[5] = P.t[1]; gt_pose[6] = P.t[2];
  }
  // Azimuth buckets (1 deg, world frame) of candidate boxes, widened by the box's angular
  // radius plus 3 deg to absorb roll/pitch.
  const int NB = 360;
  std::vector<std::vector<int>> bucket(NB);
  const double maxr = cfg->max_cast_range;
  for (size_t bi = 0; bi < g_world.boxes.size(); ++bi) {
    const Box& b = g_world.boxes[bi];
    double cx = 0.5 * (b.lo[0] + b.hi[0]) - P.t[0], cy = 0.5 * (b.lo[1] + b.hi[1]) - P.t[1];
    double rad = 0.5 * std::sqrt((b.hi[0] - b.lo[0]) * (b.hi[0] - b.lo[0]) +
                                 (b.hi[1] - b.lo[1]) * (b.hi[1] - b.lo[1]));
    double dist = std::sqrt(cx * cx + cy * cy);
    if (dist - rad > maxr) continue;
    if (dist <= rad + 0.5) { for (int k = 0; k < NB; ++k) bucket[k].push_back((int)bi); continue; }
    double az = std::atan2(cy, cx) * 180.0 / M_PI;
    double half = std::asin(std::fmin(1.0, rad / dist)) * 180.0 / M_PI + 3.0;
    int k0 = (int)std::floor(az - half), k1 = (int)std::floor(az + half);
    for (int k = k0; k <= k1; ++k) bucket[((k % NB) + NB) % NB].push_back((int)bi);
  }
  const uint64_t seed = 1000ull * (uint64_t)stream + (uint64_t)scan;
  const float nanf_ = std::numeric_limits<float>::quiet_NaN();
  // (a few threads: W is ~2 000 columns; the default team of a 256-thread host costs more in fork / join than the loop takes —
  //  0.12 s per 16 x 1800 scan on the GPU box's EPYC against 3 ms with 16 threads, which dominated the GPU test suite's run time)
  int nth = 1;
#ifdef _OPENMP
  nth = omp_get_max_threads();
  if (nth > 16) nth = 16;
  if (nth < 1) nth = 1;
#endif
#pragma omp parallel for schedule(static) num_threads(nth)
  for (int c = 0; c < W; ++c) {
    double phi = 2.0 * M_PI * c / W;
    double cphi = std::cos(phi), sphi = std::sin(phi);
    for (int r = 0; r < H; ++r) {
      double th = el[r] * M_PI / 180.0;
      double ds[3] = {std::cos(th) * cphi, std::cos(th) * sphi, std::sin(th)};
      double dw[3] = {P.R[0] * ds[0] + P.R[1] * ds[1] + P.R[2] * ds[2],
                      P.R[3] * ds[0] + P.R[4] * ds[1] + P.R[5] * ds[2],
                      P.R[6] * ds[0] + P.R[7] * ds[1] + P.R[8] * ds[2]};
      double best = std::numeric_limits<double>::infinity();
      if (dw[2] < -1e-9) {  // ground plane z = -1.7
        double tg = (-1.7 - P.t[2]) / dw[2];
        if (tg > 0) best = tg;
      }
      double azw = std::atan2(dw[1], dw[0]) * 180.0 / M_PI;
      int bk = (((int)std::floor(azw)) % NB + NB) % NB;
      const std::vector<int>& cand = bucket[bk];
      for (size_t i = 0; i < cand.size(); ++i) {
        double t = ray_box(P.t, dw, g_world.boxes[cand[i]]);
        if (t > 0 && t < best) best = t;
      }
      size_t idx = (cfg->lidar_type == 0) ? ((size_t)c * H + r) : ((size_t)r * W + c);
      float* o = xyzi + 4 * idx;
      if (!(best < maxr)) {
        o[0] = nanf_; o[1] = nanf_; o[2] = nanf_;
      } else {
        uint64_t h = splitmix64(seed * 0x100000001B3ull + (uint64_t)r * 65537ull + (uint64_t)c);
        double u1 = u01(h), u2 = u01(splitmix64(h));
        double g = std::sqrt(-2.0 * std::log(u1)) * std::cos(2.0 * M_PI * u2);
        double rng = best + cfg->noise_sigma * g;
        o[0] = (float)(ds[0] * rng); o[1] = (float)(ds[1] * rng); o[2] = (float)(ds[2] * rng);
      }
      o[3] = (float)(r * 10000 + c);
    }
  }
  return 0;
}

int synth_num_boxes(void) { return g_world.built ? (int)g_world.boxes.size() : 0; }

}  // extern "C"
