#define HGS_REAL float
#include "launch_fused_impl.hpp"
