// fast_cols.hpp -- specialised output kernel (hot kernel #2): half-complex -> real inverse
// transform along h of T columns per tile, straight into the caller's maps.
//
// Same job as cols_c2r_body (kernels_body.hpp) with the transform (M = R1*R2*R3 complex points,
// Lh = 2M = FFT_H), the tile width T and the thread count fixed at compile time, and a
// PERSISTENT workgroup: one workgroup per CU owns the whole LDS (T columns of M+1 bins plus all
// tables), loops over tiles (kernel n, column tile) and software-pipelines them -- the gather of
// the next tile is issued into registers before the current tile is transformed and lands in
// LDS after its last stage has been read.  No run-time integer division, twiddles from LDS
// tables (stage 2) or one table entry + power chain (stage 1), stage 3 on 16-byte LDS accesses.
//
// The rows of the intermediate Y arrive in the generic kernels' h-frequency order; `rowoff`
// (precomputed per LDS position) redirects the gather, so this kernel is independent of the
// radix sequence the producers use.
#pragma once
#include "butterflies.hpp"
#include "fast_rows.hpp"  // power_chain, c32x2
#include "fc_common.hpp"
#include "fc_instrument.hpp"

namespace fc {

// LDS image of a column: the R1 stage-1 blocks of m1 = R2 * R3 cells lie S1 = m1 + pad cells apart (row kernels: RowCfg::S1,
// fast_rows.hpp).  m1 cells are a multiple of all banks for most lengths (2112 = 6.16.22: 352 cells = 704 dwords), so the lanes
// of a stage-2 access on the two sides of a block boundary -- b runs over R3 cells, then the next block -- collide, and so do
// the landing stores of consecutive bin pairs (their positions differ in the FIRST digit: m1 cells apart).  With S1 = R3 (mod
// 32 cells) the bank sequence continues across the boundary.  Stage 3 reads 16-byte slots at a stride of R3 / 2 slots per run;
// the pad shifts the runs of every other block by pad / 2 slots against their neighbours in the same 16-lane access group, so
// the threads of odd blocks take their runs rotated by `rot` (the assignment of runs to threads is free: a stage-3 butterfly
// works in place).  Both per configuration from tools/lds_bank_model.py, which counts the LDS cycles of every access of a tile
// with the bank rules of the hardware (SQ_LDS_BANK_CONFLICT of the cfg3 kernel: model 27.4 %, measured 29.2 % of the LDS
// cycles without the pad; model 6 % with it).  FC_COLS_NO_BLOCK_PAD = 1 (diagnostic builds): the dense image.
// X(M, R1, R2, R3, T, pad, rot): from `tools/lds_bank_model.py --search` (it lists a pad for every configuration of fast_paths.hpp:
// 7-35 % fewer modelled LDS cycles per tile).  Only the configurations whose output kernel MEASURED faster on the padded image are
// listed -- the kernel is bound by its memory stream, not by the LDS array, and a larger image is not free (same-box A/Bs of 4-6
// fresh processes per variant, profiles/r05u_cols_block_padding_all_configs.txt): M = 2112 (cfg3 / cfg4) -0.6 %, 1536 -1 %, 1280
// -0.5 ... -1 %, 576 -1.7 ... -2.7 %, 384 -3 %; within +-0.5 % at M = 144, 288, 672, 880, 960, 1056, 1760, 1920, 2304, 2816, 3072, 3840;
// SLOWER at M = 768 (+2 %), 1152 (+1 %), 1408 (+2.4 %).  A configuration that is not listed runs on the dense image (pad 0, rot 0).
#define FC_COL_LAYOUTS(X)      \
    X(2112, 6, 16, 22, 8, 22, 15) /* LDS cycles per tile 7896 -> 6102 (SQ_LDS_IDX_ACTIVE -17 %, SQ_LDS_BANK_CONFLICT -64 %) */ \
    X(1536, 8, 16, 12, 8, 12, 0)  /* 7801 -> 5565 */ \
    X(1280, 8, 16, 10, 8, 10, 15) /* 5609 -> 3687 */ \
    X(576, 6, 8, 12, 16, 14, 0)   /* 6739 -> 5355 */ \
    X(384, 4, 8, 12, 16, 6, 0)    /* 4567 -> 3631 */

struct ColLayout { int pad, rot; };
constexpr ColLayout col_layout(int M, int R1, int R2, int R3, int T) {
    if (FC_COLS_NO_BLOCK_PAD) return {0, 0};
#define FC_X(MM, A, B, C, TT, PP, RR) \
    if (M == MM && R1 == A && R2 == B && R3 == C && T == TT) return {PP, RR};
    FC_COL_LAYOUTS(FC_X)
#undef FC_X
    return {0, 0};
}

template <int M_, int R1_, int R2_, int R3_, int T_, int NT_>
struct ColCfg {
    static constexpr int M = M_, R1 = R1_, R2 = R2_, R3 = R3_, T = T_, NT = NT_;
    static constexpr int m1 = M / R1;            // stage-1 sub-length (= R2*R3)
    static constexpr int NB1 = m1, NB2 = R1 * R3, NB3 = R1 * R2;  // butterflies per column
    static constexpr int PAD = col_layout(M_, R1_, R2_, R3_, T_).pad, ROT = col_layout(M_, R1_, R2_, R3_, T_).rot;
    static constexpr int S1 = m1 + PAD;          // LDS distance of two stage-1 blocks
    static constexpr int MP = R1 * S1;           // cells of a column's image; the Nyquist slot sits at MP
    static constexpr int LP = ((MP + 1 + 13) / 16) * 16 + 2;     // column pitch: >= MP+1, == 2 mod 16
    // LDS cell of position p of the dense sequence (p = M: the Nyquist slot)
    static FC_HD constexpr int cell(int p) { return PAD ? p + (p / m1) * PAD : p; }
    // start of stage-3 run q (q = block * R2 + c: cells [c * R3, (c + 1) * R3) of block `block`), as thread q takes it
    static FC_HD constexpr int run_of_thread(int q) {
        if constexpr (PAD == 0 && ROT == 0) return q * R3;
        else {
            const int blk = q / R2;
            int c = q - blk * R2;
            if constexpr (ROT != 0) {
                if (blk & 1) c = c + ROT >= R2 ? c + ROT - R2 : c + ROT;
            }
            return blk * S1 + c * R3;
        }
    }
    static constexpr int UPT = (M * T) / (2 * NT);               // 16-byte gather units per thread
    static constexpr int NPAIR = M / 2 - 1;                      // ordinary pairs per column
    static constexpr int RNDP = (NPAIR * T + NT - 1) / NT;
    static constexpr int RND2 = (NB2 * T + NT - 1) / NT;
    static constexpr int RND1 = (NB1 * T + NT - 1) / NT;
    static constexpr int T2N = (R2 - 1) * R3;
    // LDS image (c32 units): columns | stage-2 twiddles | stage-1 base twiddles | pair table
    static constexpr int OFF_T2 = T * LP;
    static constexpr int OFF_T1 = OFF_T2 + T2N;
    // pair table, compact: positions packed as a | b << 16 (one dword per pair), the twiddle
    // w_N^k rebuilt as WH[k >> 5] * WL[k & 31] from two tiny tables
    static constexpr int NPE = M / 2 + 1;                                 // pair entries (incl. DC, middle)
    static constexpr int NWH = (NPE + 31) / 32;
    static constexpr int OFF_WH = OFF_T1 + m1;
    static constexpr int OFF_WL = OFF_WH + NWH;
    static constexpr int OFF_PAIR = OFF_WL + 32;                          // NPE dwords = (NPE + 1) / 2 c32
    static constexpr int OFF_QUEUE = OFF_PAIR + (NPE + 1) / 2;            // 4 ints of the dynamic tile queue (TileQueue below)
    static constexpr int LDS_ELEMS = OFF_QUEUE + 2;
    static_assert(R1 * R2 * R3 == M, "radices must multiply to M");
    static_assert(M % 2 == 0 && T % 2 == 0, "even M and T");
    static_assert(NB3 * T == NT, "one stage-3 butterfly per thread");
    static_assert((M * T) % (2 * NT) == 0, "gather units must divide evenly");
    static_assert(R3 % 2 == 0, "stage-3 runs are read 16 bytes at a time");
    static_assert(LP >= MP + 1 && LP % 16 == 2, "column pitch");
    static_assert(PAD % 2 == 0 && PAD >= 0 && ROT >= 0 && ROT < R2, "stage-3 runs must stay 16-byte aligned");
    static_assert(MP < 65536, "pair positions are packed in 16 bits");
    static_assert(LDS_ELEMS * 8 <= 160 * 1024, "LDS budget");
};

// Dynamic tile queue of the persistent column kernels (plan option "dynamic_tiles").  The static deal -- workgroup b
// owns tiles b, b + grid, ... -- assumes that every workgroup of the grid starts at once and runs at the same speed.
// Beside another kernel (the RCCL broadcast of the next step's image spectrum on the ranks of a multi-GPU run:
// src/cudaConvFFTDataStreams.cu:279-289,338-447 is the intent) that does not hold: a workgroup here needs nearly a whole CU's
// LDS, so one whose CU is taken starts only when the other kernel has left, and its 1 / grid of the launch's tiles waits with it.
// With the queue a workgroup takes its next tile when it is ready for one: eight counters (one 128-byte line each), one
// per XCD; ticket k of XCD v is tile ((k >> shift) * 8 + v) << shift | (k & mask) -- chunks of 2^shift consecutive tiles
// dealt to the XCDs in turn, so the workgroups of an XCD still walk a run of adjacent tiles together through one L2 (adjacent
// tiles share the 128-byte lines of the intermediate).  A workgroup whose XCD has run out takes tickets of the others (one
// look at the seven other counters, then a take from the first that has tiles left), so a slower XCD is helped out at the
// end.  The ticket of the tile after next is requested at the start of a tile and read after its last stage: the software
// pipeline (gather of the next tile behind the transform of this one) is as in the static deal.  The counters are zero
// between launches: the plan zeroes them once, and the workgroup that leaves a launch LAST (an exit count beside the
// counters) zeroes them again -- no memset command per launch (a hipMemsetAsync ahead of every launch serialised the
// plan's stream against the side stream's kernel in most runs: profiles/r05a_contention_ab_memset_per_launch.txt).
constexpr int FC_QUEUE_STRIDE = 32;                    // ints between two counters: a 128-byte line each
constexpr int FC_QUEUE_COUNTERS = 10;                  // 8 XCD counters of the output kernel + 2 plain counters of the forward column kernels
constexpr int FC_QUEUE_EXIT = 16;                      // word of a counter's line that counts the workgroups that have left the launch
constexpr int FC_QUEUE_WORDS = FC_QUEUE_COUNTERS * FC_QUEUE_STRIDE;
FC_HD int queue_tile(int v, int k, int shift) { return ((((k >> shift) << 3) + v) << shift) + (k & ((1 << shift) - 1)); }
// thread 0 only: a tile of another XCD's queue, or n if none has any left.  Counters only grow, so a counter seen
// exhausted stays exhausted and one pass is complete; one seen with tiles left may have lost them since (the take says).
FC_HD int queue_steal(int* q, int shift, int home, int n) {
    int seen[8];
    static_for<1, 8>([&](auto i_) { seen[decltype(i_)::value] = FC_QUEUE_PEEK(q + ((home + decltype(i_)::value) & 7) * FC_QUEUE_STRIDE); });
    for (int i = 1; i < 8; i++) {
        const int v = (home + i) & 7;
        if (queue_tile(v, seen[i], shift) >= n) continue;
        const int tl = queue_tile(v, FC_QUEUE_TAKE(q + v * FC_QUEUE_STRIDE), shift);
        if (tl < n) return tl;
    }
    return n;
}

// thread 0 of a workgroup that takes no more tiles: the last one out zeroes the `n` counters from q on for the next launch
FC_HD void queue_leave(int* q, int n, int nwg) {
    if (FC_QUEUE_TAKE(q + FC_QUEUE_EXIT) == nwg - 1) {
        for (int v = 0; v < n; v++) FC_QUEUE_PUT(q + v * FC_QUEUE_STRIDE, 0);
        FC_QUEUE_PUT(q + FC_QUEUE_EXIT, 0);
    }
}

struct FastColsArgs {
    const c32* Y;            // [n][i][y_pitch]
    size_t y_kernel_stride;
    int y_pitch;
    float* out;              // kernel n at out + n*out_kernel_stride; (h, w) at w*fft_h + h
    size_t out_kernel_stride;
    int fft_h, fft_w;        // output window: fft_h <= 2M (rows beyond it are cropped), fft_w % T == 0,
                             // every column < fft_w exists in Y
    // The window may be a sub-rectangle of the transform (overlap-save blocks of a block-wise plan, fftconv_api.cpp:
    // the first rows / columns of a block's circular result are wrapped and belong to nobody): rows [h_lo, fft_h) of
    // columns [w_first, w_first + tiles_per_kernel * T) are stored, row h of column w at out + w * out_pitch + h (the
    // host offsets `out` so that this lands in the full map).  Plain plans: h_lo = 0, w_first = 0, out_pitch = fft_h.
    int h_lo, w_first, out_pitch;   // h_lo even, w_first % T == 0, out_pitch even
    int tiles_per_kernel;    // stored columns / T
    int ntiles;              // tiles_per_kernel * kernels in this launch
    int y_tiled;             // 1: Y is tiled [w / TL][row][TL] with the rows of bins (k, M-k) adjacent (rows 2k, 2k+1;
                             //    M+2 rows per tile): merged in registers on the way into LDS.  0: row-major [i][y_pitch]
    int y_tile_elems;        // (M+2) * TL
    int y_tile_shift;        // log2(TL): TL = 16, one 128-byte line per row and tile
    const int* rowoff;       // row-major Y only, M+1 entries: Y row offset (row * y_pitch) feeding LDS position p
    const c32* tw1;          // w_M^j, j < m1
    const c32* tw2;          // stage-2 table [(c-1)*R3 + b]
    const PairEntry* pairs;  // NPE entries: [0] DC/Nyquist, [k] pair (k, M-k), [M/2] middle (w = w_N^k)
    unsigned long long* timeline;  // FC_COLS_TIMELINE builds only: per-phase wall-clock stamps of workgroup 0 (else unused)
    // SLICED launches (small transforms whose last round of tiles would leave most workgroups idle): `ntiles` covers the
    // full rounds only; the `tail_tiles` tiles from `tail_first` on are cut into 1 << slice_shift column slices each and
    // dealt one slice per workgroup.  All three are 0 otherwise.
    int tail_first, tail_tiles, slice_shift;
    // dynamic tile queue (see above): 8 counters FC_QUEUE_STRIDE ints apart, zero at launch; nullptr: the static deal.
    // Unsliced launches only.
    int* queue;
    int queue_shift;
};

template <class C>
struct ColState {
    c32x2 pre[C::UPT];   // gather of the next tile
    c32x2 pre_ny;        // ... its Nyquist row (threads < T/2)
    int off[C::UPT];     // Y row offsets (or, precombined, LDS landing positions) of this thread's gather units
    int tk;              // thread 0, dynamic tile queue: the ticket requested at the start of the tile
};

// tiled intermediate (pair-adjacent rows, merge while landing): rows of bins k and M-k come as pairs
template <class C>
struct ColPairState {
    static constexpr int NPU = (C::M / 2 + 1) * (C::T / 2);        // pair units (pair, 2 columns) per tile
    static constexpr int RNDU = (NPU + C::NT - 1) / C::NT;
    c32x2 pa[RNDU];      // row of bin k (or DC / middle)
    c32x2 pb[RNDU];      // row of bin M-k (or Nyquist / padding)
    int tk;              // thread 0, dynamic tile queue: the ticket requested at the start of the tile
};

// Bin pair a gather unit takes (tiled intermediate).  With 4-column tiles a 16-lane LDS access group lands 8 pairs x 2 column
// pairs; consecutive pairs k, k + 1 sit m1 = M / R1 positions apart in the LDS image, which is a multiple of all 32 banks for
// most of these configurations (M = 4224: 528 complex = 1056 dwords) -- an 8-way conflict on every landing store.  So inside
// every full block of 64 pairs the order is transposed 8 x 8: the pairs of one access group are then 8 apart, i.e. R3 cells
// apart in the image (44 dwords at R3 = 22: eight distinct bank pairs).  Where R3 cells are a multiple of the 32 banks
// themselves (R3 = 16) the pairs of a group are taken NB3 = R1 * R2 apart instead -- they differ in the LAST digit of their
// position and land in 8 consecutive cells -- inside superblocks of 8 * NB3 pairs, the 8 x 8 form for what is left.
// Output kernel per launch, same box (profiles/r04g_cols4_landing_order.txt): M = 4224 141.8 -> 126.1 us per map; M = 3072
// 2468 -> 2202 us per 32 maps, M = 2560 1668 -> 1568; for R3 = 22 / 24 the NB3 form is 1-4 % slower than the 8 x 8 one.
template <class C>
FC_HD int pair_of_unit(int u) {
    if constexpr (C::T == 4 && FC_COLS_PAIR_TRANSPOSE) {
        constexpr int NP = C::M / 2 + 1, SB = 8 * C::NB3;
        constexpr int FULL = ((2 * C::R3) % 32 == 0) ? (NP / SB) * SB : 0;
        if (u < FULL) {
            const int sb = u / SB, i = u - sb * SB;
            return sb * SB + (i & 7) * C::NB3 + (i >> 3);
        }
        return ((u | 63) < NP) ? ((u & ~63) | ((u & 7) << 3) | ((u >> 3) & 7)) : u;
    } else if constexpr (C::T == 8 && FC_COLS_PAIR_TRANSPOSE >= 2 && (2 * C::m1) % 32 == 0) {
        // 8-column tiles: 4 pairs x 4 column pairs per access group; where consecutive pairs collide on one bank (m1 cells a
        // multiple of the 32 banks: M = 2304, 2112 as 6.16.22, 1920, 1536, 1408, 1280) the 4 pairs of a group are taken NB3 apart
        constexpr int NP = C::M / 2 + 1, SB = 4 * C::NB3, FULL = (NP / SB) * SB;
        if (u < FULL) {
            const int sb = u / SB, i = u - sb * SB;
            return sb * SB + (i & 3) * C::NB3 + (i >> 2);
        }
        // what is left of a superblock: pairs of pairs NB3 apart; with the padded image (S1 = R3 mod 32, R3 = 2 mod 4) the four
        // pairs k, k + NB3, k + 1, k + 1 + NB3 of a group land on cells 0, 1, S1, S1 + 1 = four distinct residues mod 4 beside
        // the column pairs' 0, 4, 8, 12
        constexpr int SB2 = 2 * C::NB3, FULL2 = FULL + (C::PAD ? ((NP - FULL) / SB2) * SB2 : 0);
        if (u < FULL2) {
            const int sb = (u - FULL) / SB2, i = u - FULL - sb * SB2;
            return FULL + sb * SB2 + (i & 1) * C::NB3 + (i >> 1);
        }
        return u;
    } else {
        return u;
    }
}

// TILED: layout of the intermediate -- false: row-major [i][y_pitch] (the producer is a generic
// kernel), gathered through `rowoff` and merged by a table-driven pair pass over LDS; true: tiled
// with the rows of bins (k, M-k) adjacent: one thread gathers both rows of a pair for two columns
// and merges them in registers on the way into LDS, so the pair pass (and its barrier) disappears.
// A template parameter so that each variant carries only its own address arithmetic (the kernel
// sits right at the 168-VGPR budget of 3 waves per SIMD).
// SLICED (tiled intermediate only, instantiated for the small transforms only): after its full tiles a workgroup
// takes one column SLICE of a tile of the last, partial round -- the same phases with the lanes of the other columns
// switched off -- so that round costs a fraction of a tile time instead of a whole one (cfg2: 16 maps x 68 tiles = 1088
// tiles on 256 workgroups are 4 full rounds + 64 tiles; as 256 quarter tiles the fifth round moves a quarter of the bytes).
// DYN: tiles from the dynamic queue (g.queue; see TileQueue above) instead of the static deal -- a template parameter so that
// the static kernel carries none of the queue's scalar state (as a run-time switch it cost 6 SGPRs and 44 SGPR spills).
template <class C, bool TILED, bool SLICED = false, bool DYN = false, class Ctx>
FC_HD void fast_cols_body(Ctx& ctx, c32* lds, const FastColsArgs& g, int wg, int nwg) {
    static_assert(!SLICED || TILED, "column slices exist for the tiled intermediate only");
    static_assert(!(SLICED && DYN), "the sliced tail round is dealt statically");
    static_assert(!SLICED || (C::T & (C::T - 1)) == 0, "column slices: the tile width must be a power of two (slices of whole column pairs)");
    constexpr bool PLAND = TILED;
    constexpr int M = C::M, R1 = C::R1, R2 = C::R2, R3 = C::R3, T = C::T, NT = C::NT, LP = C::LP, m1 = C::m1;
    constexpr int T2 = T / 2;
    using State = std::conditional_t<PLAND, ColPairState<C>, ColState<C>>;
    c32* tw2 = lds + C::OFF_T2;
    c32* tw1 = lds + C::OFF_T1;
    c32* wh = lds + C::OFF_WH;
    c32* wl = lds + C::OFF_WL;
    unsigned* ppos = reinterpret_cast<unsigned*>(lds + C::OFF_PAIR);

    // Tile order.  Adjacent column tiles share every 128-byte line of Y (a tile row is 64 bytes),
    // so they should be gathered at the same time through the same L2: workgroups b and b+8 sit
    // on the same XCD (round-robin dispatch; a speed assumption only), hence workgroup
    // wg = 8*slot + xcd takes tile (iter*nwg + xcd*(nwg/8) + slot) -- each XCD walks a contiguous
    // run of tiles.  Falls back to the plain order when nwg is not a multiple of 8.
    const int per_xcd = nwg / 8;
    const int wg_x = (nwg % 8 == 0) ? (wg % 8) * per_xcd + wg / 8 : wg;
    // it-th tile of this workgroup; SLICED: behind the full rounds (g.ntiles tiles) comes this workgroup's slice of a tail
    // tile, reported as tile index >= g.ntiles (cur_lo / cur_hi below say which columns of it are this workgroup's)
    const int full_rounds = SLICED ? g.ntiles / nwg : 0;
    const int n_total = SLICED ? g.ntiles + g.tail_tiles : g.ntiles;     // tile indices below this exist
    auto tile_of = [&](int it) -> int {
        if constexpr (SLICED) {
            if (it == full_rounds) return (wg_x < (g.tail_tiles << g.slice_shift)) ? g.tail_first + (wg_x >> g.slice_shift) : n_total;
            if (it > full_rounds) return n_total;
        }
        return it * nwg + wg_x;
    };
    // columns [lo, hi) of tile it that this workgroup transforms (the whole tile unless it is the slice round)
    auto cols_lo = [&](int it) -> int { return (SLICED && it == full_rounds) ? (wg_x & ((1 << g.slice_shift) - 1)) * (T >> g.slice_shift) : 0; };
    auto cols_hi = [&](int it) -> int { return (SLICED && it == full_rounds) ? cols_lo(it) + (T >> g.slice_shift) : T; };
    // dynamic tile queue (TileQueue above; never with SLICED): qs[0] = the tile after next (written by thread 0 behind the last
    // stage of a tile, read by everybody after that phase's barrier), qs[1] = this workgroup's first tile, qs[2] = 1 once the
    // home XCD's counter has run out (then tiles come from the other XCDs' counters, without the early request)
    constexpr bool dyn = DYN;
    [[maybe_unused]] int* qs = reinterpret_cast<int*>(lds + C::OFF_QUEUE);
    [[maybe_unused]] const int home = dyn ? FC_XCC_ID(wg) : 0;
    [[maybe_unused]] int dyn_next = n_total;
    if constexpr (dyn) {
        ctx.phase([&](int t, State&) {
            if (t == 0) {
                int* qh = g.queue + home * FC_QUEUE_STRIDE;
                const int k0 = FC_QUEUE_TAKE(qh), k1 = FC_QUEUE_TAKE(qh);
                int t0 = queue_tile(home, k0 < k1 ? k0 : k1, g.queue_shift), t1 = queue_tile(home, k0 < k1 ? k1 : k0, g.queue_shift);
                int out_of_home = 0;
                if (t0 >= n_total) { out_of_home = 1; t0 = queue_steal(g.queue, g.queue_shift, home, n_total); }
                if (t1 >= n_total) { out_of_home = 1; t1 = t0 < n_total ? queue_steal(g.queue, g.queue_shift, home, n_total) : n_total; }
                qs[1] = t0; qs[0] = t1; qs[2] = out_of_home;
            }
        });
        dyn_next = FC_UNIFORM(qs[0]);
    }
    const int first_tile = dyn ? FC_UNIFORM(qs[1]) : tile_of(0);
    [[maybe_unused]] int cur_lo = cols_lo(0), cur_hi = cols_hi(0), nxt_lo = 0, nxt_hi = T;

    // part: 0 = the whole gather; 1 / 2 (mode 3) = its first / second half of rounds -- a CU cannot
    // keep a whole tile (135 KB) of loads in flight, so issuing it in one go stalls the waves in
    // the issue itself; the second half is issued one phase later, while the first drains
    auto issue_gather = [&](int t, State& st, int tile, auto part_, [[maybe_unused]] int c_lo, [[maybe_unused]] int c_hi) {
        constexpr int part = decltype(part_)::value;
        const int kernel = tile / g.tiles_per_kernel;
        const int w0 = g.w_first + (tile - kernel * g.tiles_per_kernel) * T;
        if constexpr (PLAND) {   // rows 2p, 2p+1 = bins (p, M-p); one thread takes both for two columns
            const int tw = 1 << g.y_tile_shift;
            const c32* Yt = g.Y + (size_t)kernel * g.y_kernel_stride + (size_t)(w0 >> g.y_tile_shift) * g.y_tile_elems + (w0 & (tw - 1));
            // parts: FC_COLS_SPLIT_GATHER == 1: halves (1, 2); == 2: thirds (1, 2, 3)
            constexpr int NP = (FC_COLS_SPLIT_GATHER == 2) ? 3 : 2;
            constexpr int RB = (part == 0) ? 0 : (State::RNDU * (part - 1) + NP - 1) / NP;
            constexpr int RE = (part == 0) ? State::RNDU : (part == NP ? State::RNDU : (State::RNDU * part + NP - 1) / NP);
            static_for<RB, RE>([&](auto r_) {
                constexpr int r = decltype(r_)::value;
                const int e = t + NT * r;
                bool mine = e < State::NPU;
                if constexpr (SLICED) mine = mine && 2 * (e % T2) >= c_lo && 2 * (e % T2) < c_hi;
                if (mine) {
                    const c32* pr = Yt + ((size_t)(2 * pair_of_unit<C>(e / T2)) << g.y_tile_shift) + 2 * (e % T2);
                    if constexpr (!(FC_COLS_DBG & 4)) {
                        FC_STREAM_LOAD16(st.pa[r], pr);
                        FC_STREAM_LOAD16(st.pb[r], pr + tw);
                    }
                }
            });
        } else {
            const c32* Y = g.Y + (size_t)kernel * g.y_kernel_stride + w0;
            static_for<0, C::UPT>([&](auto r_) {
                constexpr int r = decltype(r_)::value;
                const int e = t + NT * r;
                if constexpr (!(FC_COLS_DBG & 4)) st.pre[r] = *reinterpret_cast<const c32x2*>(Y + st.off[r] + 2 * (e % T2));
            });
            if (t < T2) st.pre_ny = *reinterpret_cast<const c32x2*>(Y + g.rowoff[M] + 2 * t);
        }
    };
    auto land_gather = [&](int t, State& st, [[maybe_unused]] int c_lo, [[maybe_unused]] int c_hi) {
        if constexpr (PLAND) {
            static_for<0, State::RNDU>([&](auto r_) {
                constexpr int r = decltype(r_)::value;
                const int e = t + NT * r;
                bool mine = e < State::NPU;
                if constexpr (SLICED) mine = mine && 2 * (e % T2) >= c_lo && 2 * (e % T2) < c_hi;
                if (mine) {
                    const int k = pair_of_unit<C>(e / T2), t2 = e % T2;
                    const unsigned pp = ppos[e / T2];     // the table is in unit order here (prologue)
                    const int pa = (int)(pp & 0xffffu), pb = (int)(pp >> 16);
                    c32* z0 = lds + (2 * t2) * LP;
                    c32* z1 = z0 + LP;
                    c32x2 xa = st.pa[r], xb = st.pb[r];
#if (FC_COLS_DBG & 32) && defined(__HIP_DEVICE_COMPILE__)
                    // timing experiment: the arithmetic an inverse radix-8 w-stage would add while landing
                    // (about 21 packed operations per unit of two rows x two columns)
                    for (int dd = 0; dd < 5; dd++) {
                        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(xa.a) : "v"(xb.a), "v"(xa.b));
                        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(xa.b) : "v"(xb.b), "v"(xa.a));
                        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(xb.a) : "v"(xa.a), "v"(xb.b));
                        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(xb.b) : "v"(xa.b), "v"(xb.a));
                    }
#endif
                    if (k == 0) {                     // DC + Nyquist -> packed bin 0
                        z0[pa] = mk(xa.a.x + xb.a.x, xa.a.x - xb.a.x);
                        z1[pa] = mk(xa.b.x + xb.b.x, xa.b.x - xb.b.x);
                    } else if (k == M / 2) {          // self-paired middle bin
                        z0[pa] = mk(2.f * xa.a.x, -2.f * xa.a.y);
                        z1[pa] = mk(2.f * xa.b.x, -2.f * xa.b.y);
                    } else {
                        const c32 w = cmul(wh[k >> 5], wl[k & 31]);
                        {
                            c32 S = mk(xa.a.x + xb.a.x, xa.a.y - xb.a.y), D = mk(xa.a.x - xb.a.x, xa.a.y + xb.a.y);
                            c32 G = cmulc(D, w);
                            z0[pa] = mk(S.x - G.y, S.y + G.x);
                            z0[pb] = mk(S.x + G.y, G.x - S.y);
                        }
                        {
                            c32 S = mk(xa.b.x + xb.b.x, xa.b.y - xb.b.y), D = mk(xa.b.x - xb.b.x, xa.b.y + xb.b.y);
                            c32 G = cmulc(D, w);
                            z1[pa] = mk(S.x - G.y, S.y + G.x);
                            z1[pb] = mk(S.x + G.y, G.x - S.y);
                        }
                    }
                }
            });
        } else {
        static_for<0, C::UPT>([&](auto r_) {
            constexpr int r = decltype(r_)::value;
            const int e = t + NT * r;
            const int t2 = e % T2;
            const int p = C::cell(e / T2);
            lds[(2 * t2) * LP + p] = st.pre[r].a;
            lds[(2 * t2 + 1) * LP + p] = st.pre[r].b;
        });
        if (t < T2) {
            lds[(2 * t) * LP + C::MP] = st.pre_ny.a;
            lds[(2 * t + 1) * LP + C::MP] = st.pre_ny.b;
        }
        }
    };

    // prologue: tables into LDS, gather offsets into registers, first tile
    ctx.phase([&](int t, State& st) {
        for (int i = t; i < C::T2N; i += NT) tw2[i] = g.tw2[i];
        for (int i = t; i < m1; i += NT) tw1[i] = g.tw1[i];
        // pair positions as LDS cells; tiled intermediate: in the order the landing takes them (unit u handles pair
        // pair_of_unit(u): consecutive lanes then read consecutive dwords -- in pair order the pairs of an access group are
        // NB3 apart, a multiple of the 32 banks)
        for (int i = t; i < C::NPE; i += NT) {
                const PairEntry e = g.pairs[PLAND ? pair_of_unit<C>(i) : i];
                ppos[i] = (unsigned)C::cell(e.a) | ((unsigned)C::cell(e.b) << 16);
                if constexpr (!PLAND) {
                    if ((i & 31) == 0) wh[i >> 5] = e.w;      // w^(32*hi)
                    if (i < 32) wl[i] = e.w;                  // w^lo  (entry 0 holds w^0 = 1)
                }
            }
        if constexpr (PLAND) {
            for (int i = t; i < C::NWH; i += NT) wh[i] = g.pairs[32 * i].w;
            if (t < 32) wl[t] = g.pairs[t].w;
        }
        if constexpr (!PLAND)
        static_for<0, C::UPT>([&](auto r_) {
            constexpr int r = decltype(r_)::value;
            st.off[r] = g.rowoff[(t + NT * r) / T2];
        });
        if (first_tile < n_total) issue_gather(t, st, first_tile, IC<0>{}, cur_lo, cur_hi);
    });
    // (the landing of mode 3 reads the tables written above: it needs the barrier in between)
    if (first_tile < n_total) ctx.phase([&](int t, State& st) { land_gather(t, st, cur_lo, cur_hi); });

    [[maybe_unused]] int dyn_tile = first_tile;
    for (int it = 0;; it++) {
        const int tile = dyn ? dyn_tile : tile_of(it);
        if (tile >= n_total) break;
        const int kernel = tile / g.tiles_per_kernel;
        const int w0 = g.w_first + (tile - kernel * g.tiles_per_kernel) * T;
        const int next = dyn ? dyn_next : tile_of(it + 1);
        // dynamic queue: the ticket of the tile after next is requested now (ahead of the gather loads: the memory counter is in
        // order) and read in C4, behind the wait for the gather that is there anyway
        if constexpr (dyn) if (next < n_total) ctx.phase_nosync([&](int t, State& st) {
            if (t == 0 && !qs[2]) st.tk = FC_QUEUE_TAKE(g.queue + home * FC_QUEUE_STRIDE);
        });
        if constexpr (SLICED) { cur_lo = cols_lo(it); cur_hi = cols_hi(it); nxt_lo = cols_lo(it + 1); nxt_hi = cols_hi(it + 1); }
        FC_COLS_STAMP(0);

        // C1: issue the next tile's gather (lands after C4), then merge the half spectrum of
        // this tile into the packed complex sequence, in place (table driven)
        if constexpr (PLAND) ctx.phase_nosync([&](int t, State& st) {
            if (next < n_total) issue_gather(t, st, next, IC<(PLAND && FC_COLS_SPLIT_GATHER) ? 1 : 0>{}, nxt_lo, nxt_hi);
        });
        else ctx.phase([&](int t, State& st) {
            if (next < n_total) issue_gather(t, st, next, IC<0>{}, nxt_lo, nxt_hi);
            if constexpr (!(FC_COLS_DBG & 1))
            FC_NOUNROLL
            for (int r = 0; r < C::RNDP; r++) {   // not unrolled: keeps the register footprint small
                const int idx = t + NT * r;
                if (idx < C::NPAIR * T) {
                    const int k = idx / T + 1, col = idx % T;
                    c32* z = lds + col * LP;
                    const unsigned pp = ppos[k];
                    const int pa = (int)(pp & 0xffffu), pb = (int)(pp >> 16);
                    const c32 w = cmul(wh[k >> 5], wl[k & 31]);
                    c32 xk = z[pa], xm = z[pb];
                    c32 Ssum = mk(xk.x + xm.x, xk.y - xm.y);
                    c32 D = mk(xk.x - xm.x, xk.y + xm.y);
                    c32 G = cmulc(D, w);
                    z[pa] = mk(Ssum.x - G.y, Ssum.y + G.x);
                    z[pb] = mk(Ssum.x + G.y, -Ssum.y + G.x);
                }
            }
            if (t < T) {  // DC / Nyquist
                c32* z = lds + t * LP;
                const unsigned pp = ppos[0];
                const int pa = (int)(pp & 0xffffu), pb = (int)(pp >> 16);
                float x0 = z[pa].x, xm = z[pb].x;
                z[pa] = mk(x0 + xm, x0 - xm);
            } else if (t < 2 * T) {  // middle bin
                c32* z = lds + (t - T) * LP;
                const int pa = (int)(ppos[M / 2] & 0xffffu);
                c32 x = z[pa];
                z[pa] = mk(2.f * x.x, -2.f * x.y);
            }
        });

        FC_COLS_STAMP(1);
        // C2: inverse stage 3 (radix R3 on contiguous runs), one butterfly per thread
        ctx.template phase_dbg<(FC_COLS_DBG & 2) != 0>([&](int t, State&) {
            const int col = t / C::NB3, q = t % C::NB3;
            if constexpr (SLICED) { if (col < cur_lo || col >= cur_hi) return; }
            c32* p = lds + col * LP + C::run_of_thread(q);
            c32 v[R3];
            static_for<0, R3 / 2>([&](auto h_) {
                constexpr int h = decltype(h_)::value;
                c32x2 w = *reinterpret_cast<const c32x2*>(p + 2 * h);
                v[2 * h] = w.a;
                v[2 * h + 1] = w.b;
            });
            Dft<R3, +1>::run(v);
            static_for<0, R3 / 2>([&](auto h_) {
                constexpr int h = decltype(h_)::value;
                c32x2 w;
                w.a = v[2 * h];
                w.b = v[2 * h + 1];
                *reinterpret_cast<c32x2*>(p + 2 * h) = w;
            });
        });

        FC_COLS_STAMP(2);
        if constexpr (PLAND && FC_COLS_SPLIT_GATHER) ctx.phase_nosync([&](int t, State& st) {
            if (next < n_total) issue_gather(t, st, next, IC<2>{}, nxt_lo, nxt_hi);
        });
        // C3: inverse stage 2 (radix R2, sub-length R3)
        ctx.template phase_dbg<(FC_COLS_DBG & 2) != 0>([&](int t, [[maybe_unused]] State& st) {
            static_for<0, C::RND2>([&](auto r_) {
                constexpr int r = decltype(r_)::value;
                if constexpr (PLAND && FC_COLS_SPLIT_GATHER == 2 && r == C::RND2 - 1) {   // last third of the gather: ahead of the last round
                    if (next < n_total) issue_gather(t, st, next, IC<3>{}, nxt_lo, nxt_hi);
                }
                const int idx = t + NT * r;
                bool mine2 = idx < C::NB2 * T;
                if constexpr (SLICED) mine2 = mine2 && idx / C::NB2 >= cur_lo && idx / C::NB2 < cur_hi;
                if (mine2) {
                    const int col = idx / C::NB2, u = idx % C::NB2;
                    const int c1 = u / R3, b = u % R3;
                    c32* p = lds + col * LP + c1 * C::S1 + b;
                    c32 v[R2];
                    v[0] = p[0];
                    static_for<1, R2>([&](auto c_) {
                        constexpr int c = decltype(c_)::value;
                        v[c] = cmulc(p[c * R3], tw2[(c - 1) * R3 + b]);
                    });
                    Dft<R2, +1>::run(v);
                    static_for<0, R2>([&](auto a_) {
                        constexpr int a = decltype(a_)::value;
                        p[a * R3] = v[a];
                    });
                }
            });
        });

        FC_COLS_STAMP(3);
        // C4: inverse stage 1 straight to the map: out[w][2n], out[w][2n+1] = re, im of z[n]
        float* out = g.out + (size_t)kernel * g.out_kernel_stride;
        const int pair_lo = g.h_lo >> 1;                                 // complex pairs [pair_lo, pair_lo + nout) of a column are stored
        const unsigned nout = (unsigned)((g.fft_h - g.h_lo) >> 1);
        ctx.phase([&](int t, [[maybe_unused]] State& st) {
#if !FC_COLS_NO_PREWAIT
            // the next tile's gather (issued in C1) has had two stages to arrive: take it off the
            // memory counter now, so that landing it does not wait for the stores below
            FC_WAIT_VMEM();
#endif
            FC_COLS_STAMP(4);
            if constexpr (dyn) {
                if (next < n_total && t == 0) {   // the tile after next: the ticket requested in C1, or another XCD's
                    int tl = n_total;
                    if (!qs[2]) {
                        tl = queue_tile(home, st.tk, g.queue_shift);
                        if (tl >= n_total) qs[2] = 1;
                    }
                    if (tl >= n_total) tl = queue_steal(g.queue, g.queue_shift, home, n_total);
                    qs[0] = tl;
                }
            }
            FC_NOUNROLL
            for (int r = 0; r < C::RND1; r++) {   // one butterfly at a time: the prefetched tile stays in registers
                const int idx = t + NT * r;
                bool mine1 = idx < C::NB1 * T;
                if constexpr (SLICED) mine1 = mine1 && idx / C::NB1 >= cur_lo && idx / C::NB1 < cur_hi;
                if (mine1) {
                    const int col = idx / C::NB1, j = idx % C::NB1;
                    const c32* p = lds + col * LP + j;
                    c32 pw[R1];
                    power_chain<R1>(tw1[j], pw);
                    c32 v[R1];
                    v[0] = p[0];
                    static_for<1, R1>([&](auto c_) {
                        constexpr int c = decltype(c_)::value;
                        v[c] = cmulc(p[c * C::S1], pw[c]);
                    });
                    Dft<R1, +1>::run(v);
                    c32* o = reinterpret_cast<c32*>(out + (size_t)(w0 + col) * g.out_pitch);
                    static_for<0, R1>([&](auto a_) {
                        constexpr int a = decltype(a_)::value;
                        if constexpr (FC_COLS_DBG & 8) { if (v[a].x == 1.2345e-30f) o[j + a * m1] = v[a]; }
                        else if ((unsigned)(j + a * m1 - pair_lo) < nout) FC_STREAM_STORE(&o[j + a * m1], v[a]);
                    });
                }
            }
        });
        if constexpr (dyn) { dyn_tile = next; dyn_next = next < n_total ? FC_UNIFORM(qs[0]) : n_total; }

        FC_COLS_STAMP(5);
        // C5: the prefetched tile lands in LDS
        if (next < n_total) ctx.phase([&](int t, State& st) { land_gather(t, st, nxt_lo, nxt_hi); });
        FC_COLS_STAMP(6);
    }
    if constexpr (dyn) ctx.phase_nosync([&](int t, State&) { if (t == 0) queue_leave(g.queue, 8, nwg); });
}

}  // namespace fc
