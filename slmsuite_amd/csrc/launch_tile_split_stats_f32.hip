// Single-pass MRAF column kernels with the in-pass statistics (hgs_iterate_stats): own translation unit.
#define HGS_SPLIT_STATS 1
#include "launch_tile_split_f32.hip"
