// TEST-ONLY: row configurations of group 2 of fast_paths.hpp for the host emulator (see emu_runners.hpp)
#include "emu_runners.hpp"
namespace emu {
bool fast_rows_g2(int L, int nz2, EmuFastRows& run) { return fast_rows_dispatch_group<2>(L, nz2, run); }
bool fast_rows_fwd_g2(int L, EmuFastRowsFwd& run) { return fast_rows_fwd_dispatch_group<2>(L, run); }
}  // namespace emu
