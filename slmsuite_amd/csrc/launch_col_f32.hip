#define HGS_REAL float
#include "launch_col_impl.hpp"
