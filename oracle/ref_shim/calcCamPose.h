// Stand-in for include/calcCamPose.h (OpenCV / AprilTag front end): main/calibr_simulation.cpp:6 includes it but uses
// nothing from it.  Found before the reference's header because this directory comes first on the include path.
// TEST INFRASTRUCTURE ONLY (see mini_eigen.hpp).
#pragma once
