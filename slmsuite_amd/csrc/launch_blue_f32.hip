#define HGS_REAL float
#include "launch_blue_impl.hpp"
