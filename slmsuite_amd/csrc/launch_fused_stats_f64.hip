#define HGS_REAL double
#define HGS_STATS_TU 1
#include "launch_fused_impl.hpp"
