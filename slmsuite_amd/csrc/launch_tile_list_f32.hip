// The rule-specialised tile-resident kernels walking a tile list (sparse targets rounded to whole tiles): own translation unit.
#define HGS_TILE_LISTED 1
#include "launch_tile_rule_f32.hip"
