// liodom_replay — replays a directory of KITTI-style Velodyne scans (NNNNNN.bin, float32 x y z i)
// through the GPU path with the reference's parameters and writes the reference's result files
// (poses.txt in KITTI format, *_times.txt, nfeats.txt; src/stats.cc:73-132).
//
//   liodom_replay <scan_dir> <out_dir/> [name=value ...]     e.g. scan_lines=64 prev_frames=20
// With mapping=true the liodom_mapping node (launch/liodom.launch:41-56) is replayed on the device
// with its launch-file parameters (voxel_xysize= voxel_zsize= resolution= cells_xy= cells_z=); the
// final map is written to <out_dir>map.bin (float32 x y z i).
// threads=true runs the reference's own structure instead of the fused per-scan call: the clouds are
// pushed into SharedData (lidarClb), a FeatureExtractor thread and a LaserOdometer thread work side by
// side on the same handle (src/liodom_node.cc:89-91) and hand edge clouds over through the queue — the clouds stay on the
// device (tickets; handoff=host restores host clouds).  poll_us= sets the worker loops' sleep (2000 as the reference; 0 = yield);
// the run prints its scans/s.
#include <algorithm>
#include <cstdio>
#include <dirent.h>
#include <fstream>
#include <iostream>
#include <memory>
#include <thread>

#include "liodom_host.h"

int main(int argc, char** argv) {
  if (argc < 3) { std::fprintf(stderr, "usage: %s <scan_dir> <out_dir/> [name=value ...]\n", argv[0]); return 2; }
  const std::string dir = argv[1], out = argv[2];
  std::vector<std::string> kv(argv + 3, argv + argc);
  kv.push_back("save_results=true");
  liodom::Params* params = liodom::Params::getInstance();
  params->readParams(kv);

  std::vector<std::string> files;
  if (DIR* d = opendir(dir.c_str())) {
    while (dirent* e = readdir(d)) {
      const std::string n = e->d_name;
      if (n.size() > 4 && n.substr(n.size() - 4) == ".bin") files.push_back(dir + "/" + n);
    }
    closedir(d);
  }
  std::sort(files.begin(), files.end());
  if (files.empty()) { std::fprintf(stderr, "no .bin scans in %s\n", dir.c_str()); return 1; }

  size_t max_pts = 0;
  std::vector<liodom::PointCloud> clouds(files.size());
  for (size_t i = 0; i < files.size(); i++) {
    std::ifstream f(files[i], std::ios::binary | std::ios::ate);
    const size_t bytes = (size_t)f.tellg();
    f.seekg(0);
    clouds[i].points.resize(bytes / sizeof(liodom::Point));
    f.read(reinterpret_cast<char*>(clouds[i].points.data()), (std::streamsize)(clouds[i].points.size() * sizeof(liodom::Point)));
    clouds[i].width = (uint32_t)clouds[i].points.size(); clouds[i].height = 1;
    if (params->lidar_type_ == 1) {   // organised: rows = scan_lines
      clouds[i].height = (uint32_t)params->scan_lines_;
      clouds[i].width = (uint32_t)(clouds[i].points.size() / (size_t)params->scan_lines_);
    }
    max_pts = std::max(max_pts, clouds[i].points.size());
  }
  try {
    auto eng = std::make_shared<liodom::Engine>(*params, 0, (int)max_pts, (int)(max_pts / (size_t)params->scan_lines_ + 1));
    liodom::LaserOdometer odometer(eng);
    std::unique_ptr<liodom::Map> mapper;
    if (params->mapping_) {
      double xy = 40.0, z = 50.0, res = 0.4; int cells_xy = 2, cells_z = 1;      // liodom_mapping_node.cc:115-134
      for (const std::string& a : kv) {
        const size_t eq = a.find('=');
        if (eq == std::string::npos) continue;
        const std::string k = a.substr(0, eq), v = a.substr(eq + 1);
        if (k == "voxel_xysize") xy = std::stod(v); else if (k == "voxel_zsize") z = std::stod(v);
        else if (k == "resolution") res = std::stod(v); else if (k == "cells_xy") cells_xy = std::stoi(v);
        else if (k == "cells_z") cells_z = std::stoi(v);
      }
      mapper.reset(new liodom::Map(xy, z, res));
      odometer.attachMapper(mapper.get(), cells_xy, cells_z);
    }
    std::ofstream odom_log(out + "odom.txt");      // stamp, orientation xyzw, position, twist linear, twist angular
    odom_log.precision(17);
    bool threads = false, host_handoff = false;
    int poll_us = 2000;
    for (const std::string& a : kv) {
      if (a == "threads=true" || a == "threads=1") threads = true;
      if (a == "handoff=host") host_handoff = true;
      if (a.rfind("poll_us=", 0) == 0) poll_us = std::stoi(a.substr(8));
    }
    if (threads) {
      liodom::FeatureExtractor extractor(eng);
      extractor.setDeviceHandoff(!host_handoff);
      liodom::SharedData* sdata = liodom::SharedData::getInstance();
      sdata->poll_us = poll_us;
      const auto t_begin = std::chrono::steady_clock::now();
      std::vector<liodom::OdometryMsg> msgs;
      std::atomic<bool> running{true};
      std::thread feat_thread([&] { extractor(running); });                 // liodom_node.cc:89
      std::thread odom_thread([&] { odometer(running, &msgs, nullptr); });   // liodom_node.cc:90-91
      for (size_t i = 0; i < clouds.size(); i++) {
        if (params->save_results_) liodom::Stats::getInstance()->startFrame(liodom::Clock::now());   // lidarClb :49-52
        sdata->pushPointCloud(clouds[i], 0.1 * (double)i);                 // lidarClb :54
      }
      // (msgs is appended by the odometer thread only; its size is polled until every scan is through)
      for (int spin = 0; spin < 3000000; spin++) {
        std::this_thread::sleep_for(std::chrono::microseconds(100));
        if (liodom::Stats::getInstance()->numPoses() >= clouds.size()) break;
      }
      const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
      running = false;
      feat_thread.join();
      odom_thread.join();
      for (const liodom::OdometryMsg& msg : msgs) {
        odom_log << msg.stamp;
        for (double v : msg.orientation) odom_log << ' ' << v;
        for (double v : msg.position) odom_log << ' ' << v;
        for (double v : msg.linear) odom_log << ' ' << v;
        for (double v : msg.angular) odom_log << ' ' << v;
        odom_log << '\n';
      }
      std::printf("threads: %zu scans through the extractor / odometer threads in %.3f s = %.1f scans/s (handoff=%s, poll_us=%d)\n",
                  msgs.size(), secs, (double)msgs.size() / secs, host_handoff ? "host" : "device", poll_us);
      if (msgs.size() != clouds.size()) { std::fprintf(stderr, "liodom_replay: %zu of %zu scans processed\n", msgs.size(), clouds.size()); return 1; }
    }
    for (size_t i = 0; i < clouds.size() && !threads; i++) {
      liodom_step_info_t info;
      liodom::Pose p = odometer.processScan(clouds[i], 0.1 * (double)i, &info);
      const liodom::OdometryMsg msg = odometer.publishOdom(0.1 * (double)i, p);      // ~odom / ~twist numbers
      odom_log << msg.stamp;
      for (double v : msg.orientation) odom_log << ' ' << v;
      for (double v : msg.position) odom_log << ' ' << v;
      for (double v : msg.linear) odom_log << ' ' << v;
      for (double v : msg.angular) odom_log << ' ' << v;
      odom_log << '\n';
      if (i % 50 == 0) std::printf("scan %zu: %d edges, %d matches, t = %.3f %.3f %.3f\n", i, info.n_edges, info.matches[1], p.t[0], p.t[1], p.t[2]);
    }
    liodom::Stats::getInstance()->writeResults(out);
    if (mapper) {
      odometer.attachMapper(nullptr);
      const liodom::PointCloud m = mapper->getMap();
      std::ofstream f(out + "map.bin", std::ios::binary);
      f.write(reinterpret_cast<const char*>(m.points.data()), (std::streamsize)(m.points.size() * sizeof(liodom::Point)));
      std::printf("map: %zu points\n", m.size());
    }
  } catch (const std::exception& e) {
    std::fprintf(stderr, "liodom_replay: %s\n", e.what());
    return 1;
  }
  return 0;
}
