#define HGS_REAL float
#define HGS_REAL_IS_FLOAT 1
#include "launch_blue_impl.hpp"
