// Stand-in for the ROS message sensor_msgs/LaserScan (the fields of the public message definition
// that the reference's TranScanToPoints reads, include/utilities.h:14, src/utilities.cpp:181-215).
// TEST INFRASTRUCTURE ONLY — see mini_eigen.hpp.
#pragma once
#include <vector>
namespace sensor_msgs {
struct LaserScan {
  float angle_min = 0.f, angle_max = 0.f, angle_increment = 0.f;
  float time_increment = 0.f, scan_time = 0.f;
  float range_min = 0.f, range_max = 0.f;
  std::vector<float> ranges, intensities;
};
}  // namespace sensor_msgs
