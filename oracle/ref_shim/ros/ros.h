// Stand-in for <ros/ros.h>: the three calls main/calibr_simulation.cpp:111-113,164 makes.  TEST INFRASTRUCTURE ONLY
// (see mini_eigen.hpp).
#pragma once
namespace ros {
inline void init(int&, char**, const char*) {}
struct NodeHandle {
  NodeHandle() {}
  explicit NodeHandle(const char*) {}
};
// The node's main() ends in ros::spin() and then flows off its end without a return statement (fine for the real
// `main`, undefined behaviour for the renamed function the shim calls): spin() leaves by exception instead.
struct SpinCalled {};
inline void spin() { throw SpinCalled(); }
}  // namespace ros
