#define HGS_REAL double
#include "launch_row_impl.hpp"
