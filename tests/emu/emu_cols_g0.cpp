// TEST-ONLY: column configurations of group 0 of fast_paths.hpp for the host emulator (see emu_runners.hpp)
#include "emu_runners.hpp"
namespace emu {
bool fast_cols_g0(int M, int T, EmuFastCols& run) { return fast_cols_dispatch_group<0>(M, T, run); }
bool fast_cols_fwd_g0(int M, int T, bool pruned, EmuFastColsFwd& run) { return fast_cols_fwd_dispatch_group<0>(M, T, pruned, run); }
}  // namespace emu
