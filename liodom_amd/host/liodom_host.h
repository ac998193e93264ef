// liodom_host.h — host-side C++ mirror of the reference's hot-path classes, implemented over the
// C-ABI of include/liodom_hip.h (no ROS, no PCL, no Eigen, no Ceres).
//
// Names, argument meaning and error behaviour follow the reference so that a maintainer can swap
// the bodies of the corresponding reference classes:
//   liodom::Params            include/liodom/params.h:30-70, src/params.cc:37-110
//   liodom::FeatureExtractor  include/liodom/feature_extractor.h:62-85, src/feature_extractor.cc
//   liodom::LaserOdometer     include/liodom/laser_odometry.h:79-121, src/laser_odometry.cc
//   liodom::LocalMapManager   include/liodom/laser_odometry.h:62-76 (read-only view of the device window)
//   liodom::Stats             include/liodom/stats.h:40-80, src/stats.cc (result files)
//   liodom::Map               include/liodom/map.h:93-116, src/map.cc (mapping node's map, on the device)
//   liodom::SharedData        include/liodom/shared_data.h:40-86, src/shared_data.cc (the two hand-over queues)
// Like the reference the hot-path methods return void and report problems through a log hook;
// unlike it, failures of the GPU library also raise std::runtime_error (nothing falls back to CPU).
#pragma once
#include <array>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <memory>
#include <mutex>
#include <queue>
#include <string>
#include <vector>

#include "../../include/liodom_hip.h"

namespace liodom {

typedef std::chrono::high_resolution_clock Clock;   // include/liodom/defs.h:38

// pcl::PointXYZI without padding: the packed float4 the C-ABI takes.
struct Point { float x, y, z, intensity; };
struct PointCloud {
  std::vector<Point> points;
  uint32_t width = 0, height = 1;     // organised clouds: height rows x width columns (row-major)
  size_t size() const { return points.size(); }
};

// Pose as the reference publishes it (world <- laser): quaternion [x y z w] + translation.
struct Pose { double q[4] = {0, 0, 0, 1}; double t[3] = {0, 0, 0}; std::array<double, 12> matrix34() const; };

// Same public fields as liodom::Params (include/liodom/params.h:33-49).
class Params {
 public:
  double min_range_ = 3.0;
  double max_range_ = 75.0;
  int lidar_type_ = 0;
  int scan_lines_ = 64;
  int scan_regions_ = 8;
  int edges_per_region_ = 10;
  size_t min_points_per_scan_ = 90;
  size_t local_map_size_ = 5;
  bool save_results_ = false;
  std::string results_dir_ = "~/";
  std::string fixed_frame_ = "odom";
  std::string base_frame_ = "base_link";
  std::string laser_frame_ = "";
  bool use_imu_ = false;
  bool filter_local_map_ = false;
  bool mapping_ = false;
  bool publish_tf_ = true;

  static Params* getInstance();
  // readParams(nh) of the reference reads ROS parameters; here the same names come from
  // "name=value" strings (launch-file values), unknown names are ignored like nh.param would.
  void readParams(const std::vector<std::string>& name_value_pairs);
  liodom_params_t toC() const;
};

class Stats {
 public:
  static Stats* getInstance();
  void addPose(const std::array<double, 12>& pose34);                                          // stats.cc:36
  void addFeatureExtractionTime(const Clock::time_point& start, const Clock::time_point& end);  // :40
  void addLaserOdometryTime(const Clock::time_point& start, const Clock::time_point& end);      // :45
  void addNumOfFeats(const size_t& nfeats);                                                     // :50
  void startFrame(const Clock::time_point& start);                                              // :54
  void stopFrame(const Clock::time_point& stop);                                                // :60
  void writeResults(const std::string& dir);                                                    // :73-132
  void clear();
  size_t numPoses();
 private:
  std::vector<std::array<double, 12>> poses_;
  std::atomic<size_t> n_poses_{0};
  std::vector<double> feat_extr_, laser_odom_, frame_times_;
  std::vector<size_t> num_of_features_;
  std::mutex frame_mutex_;
  std::queue<Clock::time_point> start_times_;
};

// The mutex-guarded FIFOs between the ROS callback, the extractor thread and the odometer thread
// (src/shared_data.cc:37-89).  Headers carry only the stamp here.
// One element of the feature queue: the edge cloud as published on ~edges (host copy) and — with the device-resident
// hand-off — the ticket of the copy that stayed in HBM (liodom_extract_edges_device); ticket.seq == 0: host cloud only.
struct Features {
  PointCloud edges;
  double stamp = 0;
  liodom_edge_ticket_t ticket{0, 0, 0, 0};
};
class SharedData {
 public:
  static SharedData* getInstance();
  void pushPointCloud(const PointCloud& pc_in, double stamp);          // shared_data.cc:37-42
  bool popPointCloud(PointCloud& pc_out, double& stamp);               // :44-62
  void pushFeatures(const PointCloud& feat_in, double stamp);          // :64-69
  bool popFeatures(PointCloud& feat_out, double& stamp);               // :71-89
  void pushFeatures(Features&& f);                                     // the same with the device ticket
  bool popFeatures(Features& f);
  size_t numFeatures();
  void clear();
  // Poll interval of the two worker loops in microseconds (the reference sleeps 2 ms, feature_extractor.cc:80 /
  // laser_odometry.cc:270; 0 = yield only: throughput measurements)
  std::atomic<int> poll_us{2000};
 private:
  std::mutex pc_mutex_, feat_mutex_;
  std::queue<std::pair<PointCloud, double>> pc_buf_;
  std::queue<Features> feat_buf_;
};

// Owns the GPU handle shared by the extractor and the odometer of one stream.
class Engine {
 public:
  // pose_rotation_mode: see liodom_config_t (1 = Eigen 3.3.x Transform::rotation(), the default)
  Engine(const Params& p, int device, int max_points, int max_width, int pose_rotation_mode = 1);
  ~Engine();
  liodom_handle_t* handle() const { return h_; }
  int edge_capacity() const { return edge_cap_; }
  int rotation_mode() const { return rotation_mode_; }
 private:
  liodom_handle_t* h_ = nullptr;
  int edge_cap_ = 0;
  int rotation_mode_ = 1;
};

class FeatureExtractor {
 public:
  explicit FeatureExtractor(std::shared_ptr<Engine> engine);
  // splitPointCloud + extractFeatures (feature_extractor.cc:104-254) for one cloud.
  void extractFeatures(const PointCloud& pc_in, PointCloud& pc_edges);
  // the edges of the last scan that went through LaserOdometer::processScan (the ~edges topic, :70-75)
  void lastEdges(PointCloud& pc_edges);
  // The worker loop of the extractor thread (feature_extractor.cc:42-82): pop a cloud, extract,
  // push the features; polls every 2 ms like the reference.  Runs on the extraction side of the
  // handle, concurrently with LaserOdometer::operator() (liodom_node.cc:89-91).
  void operator()(std::atomic<bool>& running);
  // Device-resident hand-off (default): the worker loop leaves every edge cloud on the device (liodom_extract_edges_device), takes
  // the host copy for ~edges from the extraction's host-mapped mirror (liodom_wait_edges) and queues the ticket with it.
  // false: liodom_extract_edges (host cloud out), as the first version of this binding did.
  void setDeviceHandoff(bool on) { device_handoff_ = on; }
 private:
  std::shared_ptr<Engine> eng_;
  Params* params;
  Stats* stats;
  bool device_handoff_ = true;
};

class LocalMapManager {
 public:
  explicit LocalMapManager(std::shared_ptr<Engine> engine) : eng_(std::move(engine)) {}
  // getLocalMap (laser_odometry.cc:62-65): window points oldest frame first; returns nframes_
  size_t getLocalMap(PointCloud& map);
 private:
  std::shared_ptr<Engine> eng_;
};

// liodom::Map (include/liodom/map.h:93-116) on the device.  Poses are 3 x 4 row-major isometries
// (Pose::matrix34()), the Eigen::Isometry3d of the reference.
class Map {
 public:
  explicit Map(const double xy_size, const double z_size, const double res, int device = 0);   // map.cc:70-81
  virtual ~Map();
  Map(const Map&) = delete;
  Map& operator=(const Map&) = delete;
  void updateMap(const PointCloud& pc_in, const std::array<double, 12>& pose);                  // :90-129
  PointCloud getMap();                                                                          // :131-139
  PointCloud getLocalMap(const std::array<double, 12>& pose, int cells_xy = 2, int cells_z = 1); // :141-189
  liodom_map_t* handle() const { return m_; }
 private:
  PointCloud fetch(int which, const double* T, int cells_xy, int cells_z);
  liodom_map_t* m_ = nullptr;
};

// The numbers LaserOdometer::publishOdom puts into nav_msgs/Odometry, geometry_msgs/TwistStamped
// and the fixed_frame -> base_frame TF (laser_odometry.cc:395-446).
struct OdometryMsg {
  std::string frame_id, child_frame_id;   // params->fixed_frame_, params->base_frame_
  double stamp = 0;
  double orientation[4] = {0, 0, 0, 1};   // x y z w, pose * laser_to_base_
  double position[3] = {0, 0, 0};
  double linear[3] = {0, 0, 0};           // delta translation / delta stamp
  double angular[3] = {0, 0, 0};          // tf RPY of the delta rotation / delta stamp
};

class LaserOdometer {
 public:
  explicit LaserOdometer(std::shared_ptr<Engine> engine);
  // publishOdom (laser_odometry.cc:395-446) for the pose returned by the last process / processScan
  OdometryMsg publishOdom(double stamp, const Pose& pose);
  // SharedData::setLocalMap (shared_data.cc:91-96), fed by mapClb (liodom_node.cc:57-64); mapping_ only
  void setLocalMap(const PointCloud& map);
  // SharedData::setLastIMUOri (shared_data.cc:107-111), fed by imuClb (liodom_node.cc:66-70); use_imu_ only
  void setLastIMUOri(const double q_xyzw[4]);
  // laser_to_base_ (TF lookup, laser_odometry.cc:110-119); identity by default
  void setLaserToBase(const std::array<double, 12>& T);
  // Zero-latency on-device replay of the liodom_mapping node (liodom_attach_mapper); mapping_ only
  void attachMapper(Map* map, int cells_xy = 2, int cells_z = 1);
  // One pass of the loop body of LaserOdometer::operator() (laser_odometry.cc:107-267).
  Pose process(const PointCloud& feats, double stamp, liodom_step_info_t* info = nullptr);
  // The same on an edge cloud the extractor left on the device (Features::ticket)
  Pose process(const Features& feats, liodom_step_info_t* info = nullptr);
  // lidarClb -> extractor -> odometer without leaving the device (one H2D copy, one result record)
  Pose processScan(const PointCloud& pc_in, double stamp, liodom_step_info_t* info = nullptr);
  // The worker loop of the odometer thread (laser_odometry.cc:100-272): pop features, process, publish.
  // `published` (optional) receives every message in order.
  void operator()(std::atomic<bool>& running, std::vector<OdometryMsg>* published = nullptr, std::vector<Pose>* poses = nullptr);
  LocalMapManager lmap_manager;
 private:
  std::shared_ptr<Engine> eng_;
  Params* params;
  Stats* stats;
  std::array<double, 12> prev_odom_{{1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0}};      // laser_odometry.h:95
  std::array<double, 12> laser_to_base_{{1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0}};  // laser_odometry.h:104
  double prev_stamp_ = 0.0;                                                     // laser_odometry.h:97
  bool published_ = false;
  // output-rate watchdog (laser_odometry.cc:239-256; state laser_odometry.h:105-111, initial values laser_odometry.cc:83-90)
  void updateFrequencies(double in_stamp_secs, double now_secs);
  double in_freqs_[5], out_freqs_[5];
  double mean_in_freq_ = 100.0, mean_out_freq_ = 100.0;
  int num_freqs_ = 0;
  double last_in_time_secs_ = 0.0, last_out_time_secs_ = 0.0;
  int freq_warnings_ = 0;
  bool init_ = false;                                                           // laser_odometry.h:94
 public:
  // mean input / output frequency over the last five scans and the number of "Output frequency too low" warnings so far
  double meanInFreq() const { return mean_in_freq_; }
  double meanOutFreq() const { return mean_out_freq_; }
  int frequencyWarnings() const { return freq_warnings_; }
};

}  // namespace liodom
