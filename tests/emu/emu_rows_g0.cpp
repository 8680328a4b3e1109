// TEST-ONLY: row configurations of group 0 of fast_paths.hpp for the host emulator (see emu_runners.hpp)
#include "emu_runners.hpp"
namespace emu {
bool fast_rows_g0(int L, int nz2, EmuFastRows& run) { return fast_rows_dispatch_group<0>(L, nz2, run); }
bool fast_rows_fwd_g0(int L, EmuFastRowsFwd& run) { return fast_rows_fwd_dispatch_group<0>(L, run); }
}  // namespace emu
