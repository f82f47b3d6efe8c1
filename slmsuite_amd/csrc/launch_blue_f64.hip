#define HGS_REAL double
#include "launch_blue_impl.hpp"
