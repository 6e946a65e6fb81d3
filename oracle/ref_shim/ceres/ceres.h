// stand-in header: see ../mini_ceres.hpp
#pragma once
#include "../mini_ceres.hpp"
