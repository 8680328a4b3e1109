// TEST-ONLY: row configurations of group 1 of fast_paths.hpp for the host emulator (see emu_runners.hpp)
#include "emu_runners.hpp"
namespace emu {
bool fast_rows_g1(int L, int nz2, EmuFastRows& run) { return fast_rows_dispatch_group<1>(L, nz2, run); }
bool fast_rows_fwd_g1(int L, EmuFastRowsFwd& run) { return fast_rows_fwd_dispatch_group<1>(L, run); }
}  // namespace emu
