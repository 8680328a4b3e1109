// TEST-ONLY: column configurations of group 1 of fast_paths.hpp for the host emulator (see emu_runners.hpp)
#include "emu_runners.hpp"
namespace emu {
bool fast_cols_g1(int M, int T, EmuFastCols& run) { return fast_cols_dispatch_group<1>(M, T, run); }
bool fast_cols_fwd_g1(int M, int T, bool pruned, EmuFastColsFwd& run) { return fast_cols_fwd_dispatch_group<1>(M, T, pruned, run); }
}  // namespace emu
