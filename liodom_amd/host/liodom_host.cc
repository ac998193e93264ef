#include "liodom_host.h"
#include "../csrc/liodom_math.h"   // odom_message: the same header the kernels use

#include <algorithm>
#include <cstring>
#include <fstream>
#include <stdexcept>
#include <thread>

namespace liodom {

namespace {
void check(int rc, const char* what) {
  if (rc != LIODOM_OK) throw std::runtime_error(std::string(what) + ": " + liodom_last_error());
}
}  // namespace

std::array<double, 12> Pose::matrix34() const {
  // Eigen::Quaterniond::toRotationMatrix + translation (laser_odometry.cc:225-227)
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x;
  const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
  return {1 - (tyy + tzz), txy - twz, txz + twy, t[0],
          txy + twz, 1 - (txx + tzz), tyz - twx, t[1],
          txz - twy, tyz + twx, 1 - (txx + tyy), t[2]};
}

Params* Params::getInstance() { static Params inst; return &inst; }

void Params::readParams(const std::vector<std::string>& kv) {
  for (const std::string& s : kv) {
    const size_t eq = s.find('=');
    if (eq == std::string::npos) continue;
    const std::string k = s.substr(0, eq), v = s.substr(eq + 1);
    auto b = [&](const std::string& x) { return x == "true" || x == "1" || x == "True"; };
    if (k == "min_range") min_range_ = std::stod(v);                 // params.cc:40
    else if (k == "max_range") max_range_ = std::stod(v);            // :44
    else if (k == "lidar_type") lidar_type_ = std::stoi(v);          // :48
    else if (k == "scan_lines") scan_lines_ = std::stoi(v);          // :52
    else if (k == "scan_regions") scan_regions_ = std::stoi(v);      // :56
    else if (k == "edges_per_region") edges_per_region_ = std::stoi(v);   // :60
    else if (k == "save_results") save_results_ = b(v);              // :66
    else if (k == "save_results_dir") results_dir_ = v;              // :70
    else if (k == "fixed_frame") fixed_frame_ = v;                   // :74
    else if (k == "base_frame") base_frame_ = v;                     // :78
    else if (k == "laser_frame") laser_frame_ = v;                   // :82
    else if (k == "prev_frames") local_map_size_ = (size_t)std::stoi(v);  // :90-93
    else if (k == "use_imu") use_imu_ = b(v);                        // :96
    else if (k == "filter_local_map") filter_local_map_ = b(v);      // :100
    else if (k == "mapping") mapping_ = b(v);                        // :104
    else if (k == "publish_tf") publish_tf_ = b(v);                  // :108
  }
  min_points_per_scan_ = (size_t)(scan_regions_ * edges_per_region_ + 10);   // :63
}

liodom_params_t Params::toC() const {
  liodom_params_t c;
  liodom_params_default(&c);
  c.min_range = min_range_; c.max_range = max_range_;
  c.lidar_type = lidar_type_; c.scan_lines = scan_lines_;
  c.scan_regions = scan_regions_; c.edges_per_region = edges_per_region_;
  c.min_points_per_scan = min_points_per_scan_; c.local_map_size = local_map_size_;
  c.save_results = save_results_;
  std::strncpy(c.results_dir, results_dir_.c_str(), sizeof(c.results_dir) - 1);
  std::strncpy(c.fixed_frame, fixed_frame_.c_str(), sizeof(c.fixed_frame) - 1);
  std::strncpy(c.base_frame, base_frame_.c_str(), sizeof(c.base_frame) - 1);
  std::strncpy(c.laser_frame, laser_frame_.c_str(), sizeof(c.laser_frame) - 1);
  c.use_imu = use_imu_; c.filter_local_map = filter_local_map_; c.mapping = mapping_; c.publish_tf = publish_tf_;
  return c;
}

Stats* Stats::getInstance() { static Stats inst; return &inst; }
void Stats::addPose(const std::array<double, 12>& p) { poses_.push_back(p); n_poses_.fetch_add(1, std::memory_order_release); }
size_t Stats::numPoses() { return n_poses_.load(std::memory_order_acquire); }
void Stats::addFeatureExtractionTime(const Clock::time_point& s, const Clock::time_point& e) {
  feat_extr_.push_back((double)std::chrono::duration_cast<std::chrono::milliseconds>(e - s).count());   // whole ms, stats.cc:42
}
void Stats::addLaserOdometryTime(const Clock::time_point& s, const Clock::time_point& e) {
  laser_odom_.push_back((double)std::chrono::duration_cast<std::chrono::milliseconds>(e - s).count());
}
void Stats::addNumOfFeats(const size_t& n) { num_of_features_.push_back(n); }
void Stats::startFrame(const Clock::time_point& s) { std::lock_guard<std::mutex> l(frame_mutex_); start_times_.push(s); }
void Stats::stopFrame(const Clock::time_point& stop) {
  std::lock_guard<std::mutex> l(frame_mutex_);
  if (!start_times_.empty()) {
    const Clock::time_point start = start_times_.front();
    start_times_.pop();
    frame_times_.push_back((double)std::chrono::duration_cast<std::chrono::milliseconds>(stop - start).count());
  }
}
void Stats::clear() { n_poses_ = 0; poses_.clear(); feat_extr_.clear(); laser_odom_.clear(); frame_times_.clear(); num_of_features_.clear(); }

// File formats of Stats::writeResults (stats.cc:73-132): poses.txt = KITTI rows of the top 3x4 of
// the pose with default ostream precision; the other files one value per line.
void Stats::writeResults(const std::string& dir) {
  std::ofstream poses((dir + "poses.txt").c_str(), std::ios::out | std::ios::trunc);
  for (const auto& p : poses_) {
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 4; j++) {
        poses << p[i * 4 + j];
        if (j != 3) poses << " ";
        else if (i == 2) poses << std::endl;
        else poses << " ";
      }
  }
  auto dump = [&](const char* name, const std::vector<double>& v) {
    std::ofstream f((dir + name).c_str(), std::ios::out | std::ios::trunc);
    for (double x : v) f << x << std::endl;
  };
  dump("feat_ext_times.txt", feat_extr_);
  dump("laser_odom_times.txt", laser_odom_);
  {
    std::ofstream f((dir + "nfeats.txt").c_str(), std::ios::out | std::ios::trunc);
    for (size_t x : num_of_features_) f << x << std::endl;
  }
  dump("frame_times.txt", frame_times_);
}

SharedData* SharedData::getInstance() { static SharedData inst; return &inst; }
void SharedData::pushPointCloud(const PointCloud& pc_in, double stamp) {
  std::lock_guard<std::mutex> lk(pc_mutex_);
  pc_buf_.push(std::make_pair(pc_in, stamp));
}
bool SharedData::popPointCloud(PointCloud& pc_out, double& stamp) {
  std::lock_guard<std::mutex> lk(pc_mutex_);
  if (pc_buf_.empty()) return false;
  pc_out = std::move(pc_buf_.front().first); stamp = pc_buf_.front().second;
  pc_buf_.pop();
  return true;
}
void SharedData::pushFeatures(const PointCloud& feat_in, double stamp) {
  Features f; f.edges = feat_in; f.stamp = stamp;
  pushFeatures(std::move(f));
}
bool SharedData::popFeatures(PointCloud& feat_out, double& stamp) {
  Features f;
  if (!popFeatures(f)) return false;
  feat_out = std::move(f.edges); stamp = f.stamp;
  return true;
}
void SharedData::pushFeatures(Features&& f) {
  std::lock_guard<std::mutex> lk(feat_mutex_);
  feat_buf_.push(std::move(f));
}
bool SharedData::popFeatures(Features& f) {
  std::lock_guard<std::mutex> lk(feat_mutex_);
  if (feat_buf_.empty()) return false;
  f = std::move(feat_buf_.front());
  feat_buf_.pop();
  return true;
}
size_t SharedData::numFeatures() { std::lock_guard<std::mutex> lk(feat_mutex_); return feat_buf_.size(); }
void SharedData::clear() {
  std::lock_guard<std::mutex> a(pc_mutex_), b(feat_mutex_);
  pc_buf_ = {}; feat_buf_ = {};
}

Engine::Engine(const Params& p, int device, int max_points, int max_width, int pose_rotation_mode) {
  liodom_params_t cp = p.toC();
  liodom_config_t cfg;
  liodom_config_default(&cfg);
  cfg.device = device; cfg.n_streams = 1; cfg.max_points = max_points; cfg.max_width = max_width;
  cfg.pose_rotation_mode = pose_rotation_mode; rotation_mode_ = pose_rotation_mode != 0 ? 1 : 0;
  check(liodom_create(&cp, &cfg, &h_), "liodom_create");
  edge_cap_ = p.scan_lines_ * p.scan_regions_ * (p.edges_per_region_ + 1) + 64;
}
Engine::~Engine() { liodom_destroy(h_); }

FeatureExtractor::FeatureExtractor(std::shared_ptr<Engine> e)
    : eng_(std::move(e)), params(Params::getInstance()), stats(Stats::getInstance()) {}

void FeatureExtractor::extractFeatures(const PointCloud& pc_in, PointCloud& pc_edges) {
  const auto start_t = Clock::now();
  pc_edges.points.resize((size_t)eng_->edge_capacity());
  int n = 0;
  check(liodom_extract_edges(eng_->handle(), 0, reinterpret_cast<const float*>(pc_in.points.data()),
                             (int64_t)pc_in.size(), (int)pc_in.height, (int)pc_in.width,
                             reinterpret_cast<float*>(pc_edges.points.data()), nullptr, nullptr, nullptr,
                             eng_->edge_capacity(), &n), "liodom_extract_edges");
  pc_edges.points.resize((size_t)n);
  pc_edges.width = (uint32_t)n; pc_edges.height = 1;
  if (params->save_results_) {                                        // feature_extractor.cc:65-68
    stats->addFeatureExtractionTime(start_t, Clock::now());
    stats->addNumOfFeats(pc_edges.size());
  }
}

static void worker_pause(SharedData* sdata) {                          // feature_extractor.cc:80 / laser_odometry.cc:270
  const int us = sdata->poll_us.load(std::memory_order_relaxed);
  if (us > 0) std::this_thread::sleep_for(std::chrono::microseconds(us));
  else std::this_thread::yield();
}

void FeatureExtractor::operator()(std::atomic<bool>& running) {
  SharedData* sdata = SharedData::getInstance();
  PointCloud pc_curr;
  double stamp = 0;
  bool have = false;             // a cloud popped while every hand-off slot was taken: retried, the reference's queue is unbounded
  while (running) {                                                   // feature_extractor.cc:46
    if (have || sdata->popPointCloud(pc_curr, stamp)) {               // :49
      if (!device_handoff_) {
        PointCloud pc_edges;
        extractFeatures(pc_curr, pc_edges);                           // :53-59
        sdata->pushFeatures(pc_edges, stamp);                         // :77
        have = false;
      } else {
        const auto start_t = Clock::now();
        Features f;
        f.stamp = stamp;
        const int rc = liodom_extract_edges_device(eng_->handle(), 0, reinterpret_cast<const float*>(pc_curr.points.data()),
                                                   (int64_t)pc_curr.size(), (int)pc_curr.height, (int)pc_curr.width, &f.ticket);
        if (rc == LIODOM_ERR_BUSY) { have = true; worker_pause(sdata); continue; }
        check(rc, "liodom_extract_edges_device");
        have = false;
        // the cloud for ~edges (:70-75) out of the extraction's host-mapped mirror
        f.edges.points.resize((size_t)eng_->edge_capacity());
        int n = 0;
        check(liodom_wait_edges(eng_->handle(), &f.ticket, reinterpret_cast<float*>(f.edges.points.data()), nullptr, nullptr, nullptr,
                                eng_->edge_capacity(), &n), "liodom_wait_edges");
        f.edges.points.resize((size_t)n);
        f.edges.width = (uint32_t)n; f.edges.height = 1;
        if (params->save_results_) {                                  // :65-68
          stats->addFeatureExtractionTime(start_t, Clock::now());
          stats->addNumOfFeats(f.edges.size());
        }
        sdata->pushFeatures(std::move(f));                            // :77
      }
    }
    worker_pause(sdata);                                              // :80
  }
}

void FeatureExtractor::lastEdges(PointCloud& pc_edges) {
  pc_edges.points.resize((size_t)eng_->edge_capacity());
  int n = 0;
  check(liodom_get_edges(eng_->handle(), 0, reinterpret_cast<float*>(pc_edges.points.data()), nullptr, nullptr, nullptr,
                         eng_->edge_capacity(), &n), "liodom_get_edges");
  pc_edges.points.resize((size_t)n);
  pc_edges.width = (uint32_t)n; pc_edges.height = 1;
}

size_t LocalMapManager::getLocalMap(PointCloud& map) {
  int64_t n = 0; int nf = 0;
  liodom_get_window(eng_->handle(), 0, nullptr, 0, &n, &nf);         // size query
  map.points.resize((size_t)n);
  check(liodom_get_window(eng_->handle(), 0, reinterpret_cast<float*>(map.points.data()), n, &n, &nf), "liodom_get_window");
  map.width = (uint32_t)n; map.height = 1;
  return (size_t)nf;
}

Map::Map(const double xy_size, const double z_size, const double res, int device) {
  liodom_map_config_t c;
  liodom_map_config_default(&c);
  c.device = device; c.voxel_xysize = xy_size; c.voxel_zsize = z_size; c.resolution = res;
  check(liodom_map_create(&c, &m_), "liodom_map_create");
}
Map::~Map() { liodom_map_destroy(m_); }

void Map::updateMap(const PointCloud& pc_in, const std::array<double, 12>& pose) {
  check(liodom_map_update(m_, reinterpret_cast<const float*>(pc_in.points.data()), (int64_t)pc_in.size(), pose.data()),
        "liodom_map_update");
}

PointCloud Map::fetch(int which, const double* T, int cells_xy, int cells_z) {
  PointCloud out;
  int64_t n = 0;
  // size query first (LIODOM_ERR_CAPACITY with the size filled in), then the copy
  if (which == 0) liodom_map_get_all(m_, nullptr, 0, &n); else liodom_map_get_local(m_, T, cells_xy, cells_z, nullptr, 0, &n);
  out.points.resize((size_t)n);
  float* dst = reinterpret_cast<float*>(out.points.data());
  if (n > 0) {
    if (which == 0) check(liodom_map_get_all(m_, dst, n, &n), "liodom_map_get_all");
    else check(liodom_map_get_local(m_, T, cells_xy, cells_z, dst, n, &n), "liodom_map_get_local");
  }
  out.width = (uint32_t)n; out.height = 1;
  return out;
}
PointCloud Map::getMap() { return fetch(0, nullptr, 0, 0); }
PointCloud Map::getLocalMap(const std::array<double, 12>& pose, int cells_xy, int cells_z) {
  return fetch(1, pose.data(), cells_xy, cells_z);
}

void LaserOdometer::setLocalMap(const PointCloud& map) {
  check(liodom_set_received_map(eng_->handle(), 0, reinterpret_cast<const float*>(map.points.data()), (int64_t)map.size()),
        "liodom_set_received_map");
}
void LaserOdometer::setLastIMUOri(const double q_xyzw[4]) {
  check(liodom_set_imu_orientation(eng_->handle(), 0, q_xyzw), "liodom_set_imu_orientation");
}
void LaserOdometer::setLaserToBase(const std::array<double, 12>& T) {
  check(liodom_set_laser_to_base(eng_->handle(), T.data()), "liodom_set_laser_to_base");
  laser_to_base_ = T;
}

OdometryMsg LaserOdometer::publishOdom(double stamp, const Pose& pose) {
  OdometryMsg msg;
  msg.frame_id = params->fixed_frame_;             // :398
  msg.child_frame_id = params->base_frame_;        // :399
  msg.stamp = stamp;                               // :400
  const std::array<double, 12> cur = pose.matrix34();
  // first frame: prev_stamp_ is set to the frame's own stamp before publishing (:125,136), so the
  // first twist is 0 / 0 = NaN exactly as the reference publishes it
  if (!published_) { prev_stamp_ = stamp; published_ = true; }
  double out[13];
  liodom_dev::odom_message(prev_odom_.data(), cur.data(), laser_to_base_.data(), stamp - prev_stamp_, eng_->rotation_mode(), out);
  std::memcpy(msg.orientation, out, sizeof(double) * 4);
  std::memcpy(msg.position, out + 4, sizeof(double) * 3);
  std::memcpy(msg.linear, out + 7, sizeof(double) * 3);
  std::memcpy(msg.angular, out + 10, sizeof(double) * 3);
  prev_odom_ = cur;                                // the reference's prev_odom_ (:149) is the pose of the previous scan here
  prev_stamp_ = stamp;                             // :266
  return msg;
}
void LaserOdometer::attachMapper(Map* map, int cells_xy, int cells_z) {
  check(liodom_attach_mapper(eng_->handle(), 0, map ? map->handle() : nullptr, cells_xy, cells_z), "liodom_attach_mapper");
}

static double wall_secs() {      // ros::Time::now().toSec() of the reference (wall clock unless /use_sim_time)
  return std::chrono::duration<double>(std::chrono::system_clock::now().time_since_epoch()).count();
}

LaserOdometer::LaserOdometer(std::shared_ptr<Engine> e)
    : lmap_manager(e), eng_(e), params(Params::getInstance()), stats(Stats::getInstance()) {
  for (int i = 0; i < 5; i++) { in_freqs_[i] = 20.0; out_freqs_[i] = 20.0; }   // laser_odometry.cc:83-86 (100 / 5)
  last_in_time_secs_ = wall_secs();                                            // :89
  last_out_time_secs_ = last_in_time_secs_;                                    // :90
}

// laser_odometry.cc:239-256: running mean over five scans of the input rate (from the message stamps) and of the output
// rate (from the wall clock); a warning when the odometer keeps up with less than 80 % of the input.
void LaserOdometer::updateFrequencies(double in_stamp_secs, double now_secs) {
  mean_in_freq_ -= in_freqs_[num_freqs_];                                                   // :240
  in_freqs_[num_freqs_] = (1.0 / (in_stamp_secs - last_in_time_secs_)) / 5.0;               // :241
  mean_in_freq_ += in_freqs_[num_freqs_];                                                   // :242
  last_in_time_secs_ = in_stamp_secs;                                                       // :243
  mean_out_freq_ -= out_freqs_[num_freqs_];                                                 // :245
  out_freqs_[num_freqs_] = (1.0 / (now_secs - last_out_time_secs_)) / 5.0;                  // :246
  mean_out_freq_ += out_freqs_[num_freqs_];                                                 // :247
  last_out_time_secs_ = now_secs;                                                           // :248
  num_freqs_ = (num_freqs_ + 1) % 5;                                                        // :250-251
  if (mean_out_freq_ < mean_in_freq_ * 0.8) {                                               // :254-256 (ROS_WARN)
    freq_warnings_++;
    std::fprintf(stderr, "[ WARN] Output frequency too low: %2.2f (in: %2.2f)\n", mean_out_freq_, mean_in_freq_);
  }
}

Pose LaserOdometer::process(const PointCloud& feats, double stamp, liodom_step_info_t* info) {
  const auto start_t = Clock::now();
  double p[7];
  check(liodom_odometry_step(eng_->handle(), 0, reinterpret_cast<const float*>(feats.points.data()),
                             (int)feats.size(), stamp, p, info), "liodom_odometry_step");
  Pose out;
  std::memcpy(out.q, p, sizeof(double) * 4); std::memcpy(out.t, p + 4, sizeof(double) * 3);
  const auto end_t = Clock::now();                                    // laser_odometry.cc:237
  if (init_) updateFrequencies(stamp, wall_secs());                   // :239-256 (the first frame takes the other branch, :108-136)
  init_ = true;                                                       // :124
  if (params->save_results_) {                                        // :259-263
    stats->addPose(out.matrix34());
    stats->addLaserOdometryTime(start_t, end_t);
    stats->stopFrame(end_t);
  }
  return out;
}

Pose LaserOdometer::process(const Features& feats, liodom_step_info_t* info) {
  if (feats.ticket.seq == 0u) return process(feats.edges, feats.stamp, info);
  const auto start_t = Clock::now();
  double p[7];
  check(liodom_odometry_step_device(eng_->handle(), &feats.ticket, feats.stamp, p, info), "liodom_odometry_step_device");
  Pose out;
  std::memcpy(out.q, p, sizeof(double) * 4); std::memcpy(out.t, p + 4, sizeof(double) * 3);
  const auto end_t = Clock::now();                                    // laser_odometry.cc:237
  if (init_) updateFrequencies(feats.stamp, wall_secs());             // :239-256
  init_ = true;
  if (params->save_results_) {                                        // :259-263
    stats->addPose(out.matrix34());
    stats->addLaserOdometryTime(start_t, end_t);
    stats->stopFrame(end_t);
  }
  return out;
}

void LaserOdometer::operator()(std::atomic<bool>& running, std::vector<OdometryMsg>* published, std::vector<Pose>* poses) {
  SharedData* sdata = SharedData::getInstance();
  while (running) {                                                   // laser_odometry.cc:102
    Features feats;
    if (sdata->popFeatures(feats)) {                                  // :107
      const Pose p = process(feats);                                  // :108-235
      const OdometryMsg msg = publishOdom(feats.stamp, p);            // :265
      if (published) published->push_back(msg);
      if (poses) poses->push_back(p);
    }
    worker_pause(sdata);                                              // :270
  }
}

}  // namespace liodom

// ---- throughput of the two-thread binding, measured in C++ (bench.py's two_thread leg; liodom_replay threads=true) ----
// An extractor thread and an odometer thread drive ONE handle through the C-ABI exactly as the patched liodom_node would
// (INTEGRATION.md §2): liodom_extract_edges_device (+ liodom_wait_edges: the ~edges cloud) on one side, a queue of tickets,
// liodom_odometry_submit_device / liodom_odometry_collect on the other (depth 1: the next ticket, if it is already queued, is
// submitted before the previous pose is collected; depth 0: liodom_odometry_step_device).  No sleeps: both threads spin-yield.
// scans: count clouds of n points (packed XYZI) `stride_floats` apart in host memory — page-locked (liodom_pin_host_buffer)
// for asynchronous uploads.  The clock runs from the submission of scan `timed_from` to the collection of the last pose.
extern "C" int liodom_host_two_thread_replay(liodom_handle_t* h, const float* scans, int64_t stride_floats, int count, int64_t n,
                                             int height, int width, int timed_from, int fetch_edges, int depth, int edge_cap,
                                             double* poses_out /*count x 7*/, double* seconds_out, int64_t* edges_total_out) {
  if (!h || !scans || count <= 0 || !poses_out) return LIODOM_ERR_INVALID_ARG;
  std::mutex qm;
  std::queue<liodom_edge_ticket_t> q;
  std::atomic<int> rc_x{0}, rc_o{0};
  std::atomic<bool> abort_all{false};
  int64_t edges_total = 0;
  std::thread tx([&] {
    std::vector<float> ebuf((size_t)std::max(1, edge_cap) * 4);
    for (int k = 0; k < count && !abort_all; k++) {
      liodom_edge_ticket_t t;
      int rc;
      while ((rc = liodom_extract_edges_device(h, 0, scans + (size_t)k * (size_t)stride_floats, n, height, width, &t)) == LIODOM_ERR_BUSY) {
        if (abort_all) return;
        std::this_thread::yield();
      }
      if (rc) { rc_x = rc; abort_all = true; return; }
      if (fetch_edges) {
        int ne = 0;
        rc = liodom_wait_edges(h, &t, ebuf.data(), nullptr, nullptr, nullptr, edge_cap, &ne);
        if (rc) { rc_x = rc; abort_all = true; return; }
        edges_total += ne;
      }
      std::lock_guard<std::mutex> lk(qm);
      q.push(t);
    }
  });
  std::chrono::steady_clock::time_point t0{}, t1{};
  std::thread to([&] {
    int submitted = 0, collected = 0;
    const int max_fly = depth ? 2 : 1;
    while (collected < count && !abort_all) {
      liodom_edge_ticket_t t;
      bool got = false;
      if (submitted < count && submitted - collected < max_fly) {
        std::lock_guard<std::mutex> lk(qm);
        if (!q.empty()) { t = q.front(); q.pop(); got = true; }
      }
      if (got) {                                     // a ticket is waiting: its odometry goes in behind the scan in flight
        if (submitted == timed_from) t0 = std::chrono::steady_clock::now();
        const int rc = liodom_odometry_submit_device(h, &t, 0.1 * submitted);
        if (rc) { rc_o = rc; abort_all = true; return; }
        submitted++;
      } else if (submitted > collected) {            // nothing (more) to submit: the oldest pose in flight is collected (and would be published)
        const int rc = liodom_odometry_collect(h, 0, poses_out + 7 * (size_t)collected, nullptr);
        if (rc) { rc_o = rc; abort_all = true; return; }
        collected++;
      } else {
        std::this_thread::yield();
      }
    }
    t1 = std::chrono::steady_clock::now();
  });
  tx.join();
  to.join();
  if (seconds_out) *seconds_out = std::chrono::duration<double>(t1 - t0).count();
  if (edges_total_out) *edges_total_out = edges_total;
  if (rc_x) return rc_x;
  return rc_o;
}

namespace liodom {

Pose LaserOdometer::processScan(const PointCloud& pc_in, double stamp, liodom_step_info_t* info) {
  const auto start_t = Clock::now();
  if (params->save_results_) stats->startFrame(start_t);              // liodom_node.cc:49-52
  double p[7];
  liodom_step_info_t local;
  check(liodom_process_scan(eng_->handle(), 0, reinterpret_cast<const float*>(pc_in.points.data()),
                            (int64_t)pc_in.size(), (int)pc_in.height, (int)pc_in.width, stamp, p, &local),
        "liodom_process_scan");
  if (info) *info = local;
  Pose out;
  std::memcpy(out.q, p, sizeof(double) * 4); std::memcpy(out.t, p + 4, sizeof(double) * 3);
  if (init_) updateFrequencies(stamp, wall_secs());                   // laser_odometry.cc:239-256
  init_ = true;
  if (params->save_results_) {
    const auto end_t = Clock::now();
    stats->addNumOfFeats((size_t)local.n_edges);
    stats->addPose(out.matrix34());
    stats->addLaserOdometryTime(start_t, end_t);
    stats->stopFrame(end_t);
  }
  return out;
}

}  // namespace liodom
