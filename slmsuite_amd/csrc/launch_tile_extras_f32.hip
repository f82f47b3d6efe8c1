#define HGS_REAL float
#define HGS_REAL_IS_FLOAT 1
#define HGS_TILE_EXTRAS_TU 1
#include "launch_fused_impl.hpp"
