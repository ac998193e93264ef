// liodom_mapping — ROS 1 transport around the device liodom::Map (SOURCE ONLY: never compiled in the
// build image).  Topics / parameters of the reference mapping node (src/liodom_mapping_node.cc:45-149).
#include <memory>
#include <string>

#include <ros/ros.h>
#include <sensor_msgs/PointCloud2.h>
#include <tf/transform_listener.h>

#include "cloud_io.h"
#include "liodom_host.h"

namespace {

struct MappingNode {
  ros::NodeHandle nh{"~"};
  std::unique_ptr<liodom::Map> mapper;
  std::unique_ptr<tf::TransformListener> listener;
  ros::Subscriber points_sub;
  ros::Publisher map_pub, local_pub;
  ros::WallTimer timer;
  ros::WallTime last_pub = ros::WallTime::now();
  std::string fixed_frame, base_frame;
  int cells_xy = 2, cells_z = 1;

  void publish_cloud(ros::Publisher& pub, const liodom::PointCloud& pc) {
    std_msgs::Header h;
    h.frame_id = fixed_frame;
    sensor_msgs::PointCloud2 msg;
    liodom_ros::to_msg(pc, h, msg);
    pub.publish(msg);
  }

  void points_cb(const sensor_msgs::PointCloud2ConstPtr& msg) {
    liodom::PointCloud cloud;
    if (!liodom_ros::from_msg(*msg, cloud)) return;
    ROS_INFO_STREAM("Received cloud with " << cloud.size() << " points.");
    tf::StampedTransform t;                                   // pose of the cloud's stamp (:53-58)
    try {
      listener->waitForTransform(fixed_frame, base_frame, msg->header.stamp, ros::Duration(5.0));
      listener->lookupTransform(fixed_frame, base_frame, msg->header.stamp, t);
    } catch (const tf::TransformException& ex) {
      ROS_ERROR("%s", ex.what());
      ros::Duration(1.0).sleep();
    }
    const tf::Matrix3x3& R = t.getBasis();
    const tf::Vector3& o = t.getOrigin();
    const std::array<double, 12> T{{R[0][0], R[0][1], R[0][2], o.x(), R[1][0], R[1][1], R[1][2], o.y(), R[2][0], R[2][1], R[2][2], o.z()}};
    try {
      mapper->updateMap(cloud, T);                                                        // :69
      if (map_pub.getNumSubscribers() > 0) publish_cloud(map_pub, mapper->getMap());      // :72-78
      if (local_pub.getNumSubscribers() > 0) publish_cloud(local_pub, mapper->getLocalMap(T, cells_xy, cells_z));   // :81-87
    } catch (const std::exception& e) {
      ROS_ERROR("%s", e.what());
    }
    last_pub = ros::WallTime::now();
  }

  void timer_cb(const ros::WallTimerEvent&) {                 // whole map again if idle for > 5 s (:92-101)
    if (map_pub.getNumSubscribers() > 0 && (ros::WallTime::now() - last_pub).toSec() > 5.0) publish_cloud(map_pub, mapper->getMap());
  }

  int run() {
    double xy, z, res;
    nh.param("voxel_xysize", xy, 40.0);
    nh.param("voxel_zsize", z, 50.0);
    nh.param("resolution", res, 0.4);
    nh.param("fixed_frame", fixed_frame, std::string("world"));
    nh.param("base_frame", base_frame, std::string("base_link"));
    nh.param("cells_xy", cells_xy, 2);
    nh.param("cells_z", cells_z, 1);
    ROS_INFO("Voxel size (XY) %.2f, (Z) %.2f, resolution %.2f, local map cells %d / %d", xy, z, res, cells_xy, cells_z);
    try {
      mapper.reset(new liodom::Map(xy, z, res));
    } catch (const std::exception& e) {
      ROS_FATAL("%s", e.what());
      return 1;
    }
    listener.reset(new tf::TransformListener());
    points_sub = nh.subscribe("points", 1, &MappingNode::points_cb, this);
    map_pub = nh.advertise<sensor_msgs::PointCloud2>("map", 1, true);
    local_pub = nh.advertise<sensor_msgs::PointCloud2>("map_local", 1, true);
    timer = nh.createWallTimer(ros::WallDuration(3.0), &MappingNode::timer_cb, this);
    ros::spin();
    return 0;
  }
};

}  // namespace

int main(int argc, char** argv) {
  ros::init(argc, argv, "liodom_mapping");
  MappingNode node;
  return node.run();
}
