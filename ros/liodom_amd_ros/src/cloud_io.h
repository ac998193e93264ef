// PointCloud2 <-> packed XYZI without PCL (field offsets read from the message).
#pragma once
#include <cstring>
#include <limits>
#include <sensor_msgs/PointCloud2.h>
#include <sensor_msgs/PointField.h>

#include "liodom_host.h"

namespace liodom_ros {

inline int field_offset(const sensor_msgs::PointCloud2& m, const char* name) {
  for (const auto& f : m.fields)
    if (f.name == name && f.datatype == sensor_msgs::PointField::FLOAT32) return (int)f.offset;
  return -1;
}

// Organised clouds keep height x width (Ouster, lidar_type 1); everything else becomes 1 x N.
inline bool from_msg(const sensor_msgs::PointCloud2& m, liodom::PointCloud& out) {
  const int ox = field_offset(m, "x"), oy = field_offset(m, "y"), oz = field_offset(m, "z");
  int oi = field_offset(m, "intensity");
  if (ox < 0 || oy < 0 || oz < 0) return false;
  const size_t n = (size_t)m.width * m.height;
  out.points.resize(n);
  out.width = m.width; out.height = m.height;
  for (size_t i = 0; i < n; i++) {
    const uint8_t* p = m.data.data() + i * m.point_step;
    liodom::Point& q = out.points[i];
    std::memcpy(&q.x, p + ox, 4); std::memcpy(&q.y, p + oy, 4); std::memcpy(&q.z, p + oz, 4);
    if (oi >= 0) std::memcpy(&q.intensity, p + oi, 4); else q.intensity = 0.f;
  }
  return true;
}

inline void to_msg(const liodom::PointCloud& pc, const std_msgs::Header& header, sensor_msgs::PointCloud2& m) {
  m.header = header;
  m.height = 1; m.width = (uint32_t)pc.size();
  m.is_bigendian = false; m.is_dense = true;
  m.point_step = 16; m.row_step = m.point_step * m.width;
  m.fields.resize(4);
  const char* names[4] = {"x", "y", "z", "intensity"};
  for (int k = 0; k < 4; k++) {
    m.fields[k].name = names[k]; m.fields[k].offset = 4 * k;
    m.fields[k].datatype = sensor_msgs::PointField::FLOAT32; m.fields[k].count = 1;
  }
  m.data.resize((size_t)m.row_step);
  if (!pc.points.empty()) std::memcpy(m.data.data(), pc.points.data(), m.data.size());
}

}  // namespace liodom_ros
