// liodom_node — ROS 1 transport around the MI355X LiODOM path (SOURCE ONLY: never compiled in the
// build image, which has no ROS).  Same private-namespace topics and parameters as the reference
// node (src/liodom_node.cc:72-110, src/params.cc:37-110, src/laser_odometry.cc:395-446).
#include <memory>
#include <string>
#include <vector>

#include <geometry_msgs/TwistStamped.h>
#include <nav_msgs/Odometry.h>
#include <ros/ros.h>
#include <sensor_msgs/Imu.h>
#include <sensor_msgs/PointCloud2.h>
#include <tf/transform_broadcaster.h>
#include <tf/transform_listener.h>

#include "cloud_io.h"
#include "liodom_host.h"

namespace {

struct Node {
  ros::NodeHandle nh{"~"};
  liodom::Params* params = liodom::Params::getInstance();
  std::shared_ptr<liodom::Engine> engine;
  std::unique_ptr<liodom::FeatureExtractor> extractor;
  std::unique_ptr<liodom::LaserOdometer> odometer;
  std::unique_ptr<liodom::Map> mapper;            // only with in_process_mapper
  ros::Publisher edges_pub, odom_pub, twist_pub;
  ros::Subscriber points_sub, map_sub, imu_sub;
  tf::TransformBroadcaster tf_broadcaster;
  tf::TransformListener tf_listener;
  bool have_laser_to_base = false;
  int max_points = 0, max_width = 0;

  // every parameter the reference reads, forwarded as name=value to the host mirror's readParams
  void read_params() {
    std::vector<std::string> kv;
    auto fwd_d = [&](const char* n, double d) { double v; nh.param(n, v, d); kv.push_back(std::string(n) + "=" + std::to_string(v)); };
    auto fwd_i = [&](const char* n, int d) { int v; nh.param(n, v, d); kv.push_back(std::string(n) + "=" + std::to_string(v)); };
    auto fwd_b = [&](const char* n, bool d) { bool v; nh.param(n, v, d); kv.push_back(std::string(n) + (v ? "=true" : "=false")); };
    auto fwd_s = [&](const char* n, const char* d) { std::string v; nh.param<std::string>(n, v, d); kv.push_back(std::string(n) + "=" + v); };
    fwd_d("min_range", 3.0); fwd_d("max_range", 75.0);
    fwd_i("lidar_type", 0); fwd_i("scan_lines", 64); fwd_i("scan_regions", 8); fwd_i("edges_per_region", 10);
    fwd_b("save_results", false); fwd_s("save_results_dir", "~/");
    fwd_s("fixed_frame", "odom"); fwd_s("base_frame", "base_link"); fwd_s("laser_frame", "");
    fwd_i("prev_frames", 5);
    fwd_b("use_imu", false); fwd_b("filter_local_map", false); fwd_b("mapping", false); fwd_b("publish_tf", true);
    params->readParams(kv);
    nh.param("max_points", max_points, 300000);          // capacities: no counterpart in the reference
    nh.param("max_width", max_width, 4096);
  }

  void lookup_laser_to_base(const std_msgs::Header& header) {
    // laser_odometry.cc:110-119: base_frame <- laser frame, once, at the first cloud
    const std::string laser = params->laser_frame_.empty() ? header.frame_id : params->laser_frame_;
    tf::StampedTransform t;
    try {
      tf_listener.waitForTransform(laser, params->base_frame_, ros::Time(0), ros::Duration(5.0));
      tf_listener.lookupTransform(laser, params->base_frame_, ros::Time(0), t);
    } catch (const tf::TransformException& ex) {
      ROS_ERROR("%s", ex.what());
      return;
    }
    const tf::Matrix3x3& R = t.getBasis();
    const tf::Vector3& o = t.getOrigin();
    std::array<double, 12> T{{R[0][0], R[0][1], R[0][2], o.x(), R[1][0], R[1][1], R[1][2], o.y(), R[2][0], R[2][1], R[2][2], o.z()}};
    odometer->setLaserToBase(T);
    have_laser_to_base = true;
  }

  void points_cb(const sensor_msgs::PointCloud2ConstPtr& msg) {
    liodom::PointCloud cloud;
    if (!liodom_ros::from_msg(*msg, cloud)) { ROS_ERROR_ONCE("~points needs float32 x, y, z fields"); return; }
    if (params->lidar_type_ != 1) { cloud.height = 1; cloud.width = (uint32_t)cloud.size(); }
    if (!have_laser_to_base) lookup_laser_to_base(msg->header);
    liodom_step_info_t info;
    liodom::Pose pose;
    try {
      pose = odometer->processScan(cloud, msg->header.stamp.toSec(), &info);
    } catch (const std::exception& e) {
      ROS_ERROR("%s", e.what());
      return;
    }
    ROS_DEBUG("Extracted edges: %d, correct matchings: %d", info.n_edges, info.matches[1]);
    if (edges_pub.getNumSubscribers() > 0) {              // feature_extractor.cc:70-75
      liodom::PointCloud edges;
      extractor->lastEdges(edges);
      sensor_msgs::PointCloud2 out;
      liodom_ros::to_msg(edges, msg->header, out);
      edges_pub.publish(out);
    }
    publish(msg->header, odometer->publishOdom(msg->header.stamp.toSec(), pose));
  }

  void publish(const std_msgs::Header& header, const liodom::OdometryMsg& m) {   // laser_odometry.cc:395-446
    nav_msgs::Odometry odom;
    odom.header.frame_id = m.frame_id; odom.child_frame_id = m.child_frame_id; odom.header.stamp = header.stamp;
    odom.pose.pose.orientation.x = m.orientation[0]; odom.pose.pose.orientation.y = m.orientation[1];
    odom.pose.pose.orientation.z = m.orientation[2]; odom.pose.pose.orientation.w = m.orientation[3];
    odom.pose.pose.position.x = m.position[0]; odom.pose.pose.position.y = m.position[1]; odom.pose.pose.position.z = m.position[2];
    odom.twist.twist.linear.x = m.linear[0]; odom.twist.twist.linear.y = m.linear[1]; odom.twist.twist.linear.z = m.linear[2];
    odom.twist.twist.angular.x = m.angular[0]; odom.twist.twist.angular.y = m.angular[1]; odom.twist.twist.angular.z = m.angular[2];
    odom_pub.publish(odom);
    geometry_msgs::TwistStamped tw;
    tw.header.frame_id = m.child_frame_id; tw.header.stamp = header.stamp; tw.twist = odom.twist.twist;
    twist_pub.publish(tw);
    if (params->publish_tf_) {
      tf::Transform t;
      t.setOrigin(tf::Vector3(m.position[0], m.position[1], m.position[2]));
      t.setRotation(tf::Quaternion(m.orientation[0], m.orientation[1], m.orientation[2], m.orientation[3]));
      tf_broadcaster.sendTransform(tf::StampedTransform(t, header.stamp, m.frame_id, m.child_frame_id));
    }
  }

  void map_cb(const sensor_msgs::PointCloud2ConstPtr& msg) {      // mapClb, liodom_node.cc:57-64
    liodom::PointCloud map;
    if (liodom_ros::from_msg(*msg, map)) odometer->setLocalMap(map);
  }
  void imu_cb(const sensor_msgs::ImuConstPtr& msg) {              // imuClb, liodom_node.cc:66-70
    const double q[4] = {msg->orientation.x, msg->orientation.y, msg->orientation.z, msg->orientation.w};
    odometer->setLastIMUOri(q);
  }

  int run() {
    read_params();
    try {
      engine = std::make_shared<liodom::Engine>(*params, 0, max_points, max_width);
      extractor.reset(new liodom::FeatureExtractor(engine));
      odometer.reset(new liodom::LaserOdometer(engine));
      bool in_process = false;
      nh.param("in_process_mapper", in_process, false);
      if (params->mapping_ && in_process) {
        double xy, z, res; int cxy, cz;
        nh.param("voxel_xysize", xy, 40.0); nh.param("voxel_zsize", z, 50.0); nh.param("resolution", res, 0.4);
        nh.param("cells_xy", cxy, 2); nh.param("cells_z", cz, 1);
        mapper.reset(new liodom::Map(xy, z, res));
        odometer->attachMapper(mapper.get(), cxy, cz);
      }
    } catch (const std::exception& e) {
      ROS_FATAL("%s", e.what());
      return 1;
    }
    edges_pub = nh.advertise<sensor_msgs::PointCloud2>("edges", 10);
    odom_pub = nh.advertise<nav_msgs::Odometry>("odom", 10);
    twist_pub = nh.advertise<geometry_msgs::TwistStamped>("twist", 10);
    points_sub = nh.subscribe("points", 1, &Node::points_cb, this);
    if (params->mapping_ && !mapper) map_sub = nh.subscribe("map", 1, &Node::map_cb, this);
    if (params->use_imu_) imu_sub = nh.subscribe("imu", 1, &Node::imu_cb, this);
    ros::spin();
    if (params->save_results_) liodom::Stats::getInstance()->writeResults(params->results_dir_);   // liodom_node.cc:112-116
    if (mapper) odometer->attachMapper(nullptr);
    return 0;
  }
};

}  // namespace

int main(int argc, char** argv) {
  ros::init(argc, argv, "liodom");
  Node node;
  return node.run();
}
