#define HGS_REAL float
#define HGS_REAL_IS_FLOAT 1
#include "launch_row_impl.hpp"
namespace hgs { int row_fpw(int N) { const int T = N / 16; return (T >= 256 ? T : 256) / T; } }
