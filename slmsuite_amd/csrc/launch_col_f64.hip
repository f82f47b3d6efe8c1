#define HGS_REAL double
#include "launch_col_impl.hpp"
