#define HGS_REAL double
#include "launch_fused_impl.hpp"
