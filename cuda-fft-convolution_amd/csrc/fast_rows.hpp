// fast_rows.hpp -- the specialised spectral-row kernel (hot kernel #1 of the path): configuration type, argument block and
// helpers.  The kernel body is fast_rows_multi_body (fast_rows_multi.hpp): a workgroup walks one or more maps (until round 4
// a separate one-map body lived here).
//
// Same job as spectral_rows_body (kernels_body.hpp) -- forward w-transform of the kernel's column
// spectrum row, product with the image spectrum row, sum over features, inverse w-transform --
// but with everything the generic kernel decides at run time fixed at compile time:
//   * the transform length L = R1*R2*R3 and its three radices (three in-place LDS stages);
//   * NT threads per row with exactly one stage-3 butterfly per thread, so the last forward
//     stage, the pointwise product and the first inverse stage run back to back IN REGISTERS
//     (one LDS round trip and one barrier fewer, and the image spectrum is consumed straight
//     from the VGPRs it was prefetched into at the start of the row);
//   * the zero padding of the kernel row is never materialised: stage 1 sees one non-zero input
//     per butterfly (pure twiddle scaling) and stage 2 reads only its first NZ2 inputs;
//   * stage-1 twiddles are one table entry + an in-register power chain, stage-2 twiddles a
//     small LDS table; no integer division by run-time values anywhere.
// The image spectrum is read in the "register order" layout (rows_fwd_body stores it that way):
// element a of stage-3 butterfly q at ((a>>1)*NB3 + q)*2 + (a&1), i.e. 16 B per lane, fully
// coalesced.
//
// The body is written as a sequence of phases over a per-thread State so the host emulator can
// run it (tests/emu) before it ever touches a GPU.
#pragma once
#include "butterflies.hpp"
#include "fc_common.hpp"
#include "fc_instrument.hpp"

namespace fc {

// RPW rows of length L are transformed side by side by one workgroup of NT threads (short rows:
// several per workgroup so that every stage still fills the lanes).
template <int L_, int R1_, int R2_, int R3_, int NT_, int RPW_ = 1>
struct RowCfg {
    static constexpr int L = L_, R1 = R1_, R2 = R2_, R3 = R3_, NT = NT_, RPW = RPW_;
    static constexpr int m1 = L / R1;        // stage-1 sub-length (= R2*R3)
    // LDS image of a row: the R1 stage-1 blocks lie S1 >= m1 cells apart.  The stage-2 accesses of a wave's lanes run over b (R3
    // consecutive cells) and then jump to the next block: with S1 = m1 that jump is a multiple of all 32 banks for most lengths
    // (528 cells at 4224 = 1056 dwords) and the lanes on both sides of it collide -- a quarter of the kernel's LDS cycles were
    // bank-conflict cycles (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE, profiles/r05s_*).  S1 = R3 (mod 16) continues the bank
    // sequence across the jump.  FC_ROWS_NO_BLOCK_PAD = 1 (diagnostic builds) restores S1 = m1.
    static constexpr int PAD1 = FC_ROWS_NO_BLOCK_PAD ? 0 : ((R3 - m1 % 16) % 16 + 16) % 16;
    static constexpr int S1 = m1 + PAD1;     // LDS distance of two stage-1 blocks
    static constexpr int LR = R1 * S1;       // LDS cells of one row
    static constexpr int NB1 = m1;           // butterflies per stage and row
    static constexpr int NB2 = R1 * R3;
    static constexpr int NB3 = R1 * R2;
    static constexpr int RND1 = (RPW * NB1 + NT - 1) / NT;
    static constexpr int RND2 = (RPW * NB2 + NT - 1) / NT;
    static constexpr int T2N = fc_tw2_pitch(R2) * R3;  // stage-2 twiddle image in LDS (fc_common.hpp: fc_tw2_fill)
    static constexpr int LDS_ELEMS = RPW * LR + T2N;  // c32
    static_assert(R1 * R2 * R3 == L, "radices must multiply to L");
    static_assert(RPW * NB3 <= NT, "one stage-3 butterfly per thread");
    static_assert(R3 % 2 == 0, "register-order layout pairs stage-3 elements");
    static_assert((R3 * 8) % 16 == 0 && (S1 * 8) % 16 == 0, "stage-3 runs must be 16-byte aligned");
};

// position of element (q, a) of the register-order layout
template <class C>
FC_HD int reg_order_index(int q, int a) {
    return ((a >> 1) * C::NB3 + q) * 2 + (a & 1);
}

struct FastRowsArgs {
    const c32* A;            // kernel column spectra [n][f][i][a_pitch]
    size_t a_kernel_stride;
    size_t a_feat_stride;
    int a_pitch;
    int kw;
    const c32* S;            // image spectrum, register-order layout, [f][i][s_pitch]
    size_t s_feat_stride;
    int s_pitch;
    c32* Y;                  // [n][i][y_pitch]
    size_t y_kernel_stride;
    int y_pitch;
    int wout;
    int F;
    const c32* tw1;          // w_L^j, j in [0, m1)
    const c32* tw2;          // stage-2 table [(c-1)*R3 + b]
    // tiled intermediate (used when the fast output kernel consumes it): element (row i, w) at
    //   (w / TL) * y_tile_elems + y_row_of[i] * TL + (w % TL),   TL = 1 << y_tile_shift columns (8 or 16)
    // so that a column tile of the output kernel is (part of) one contiguous block whose rows
    // are already in that kernel's LDS order.  y_row_of == nullptr: plain [i][y_pitch] rows.
    const int* y_row_of;
    int y_tile_elems;        // (M+1) * TL
    int y_tile_shift;        // log2(TL)
    unsigned long long* timeline;  // FC_ROWS_TIMELINE builds only: per-phase wall-clock stamps of one workgroup (else unused)
};

// p[c] = w^c, c in [1, R)
template <int R>
FC_HD void power_chain(c32 w, c32 (&p)[R]) {
    p[0] = mk(1.f, 0.f);
    if constexpr (R > 1) p[1] = w;
    static_for<2, R>([&](auto c_) {
        constexpr int c = decltype(c_)::value;
        if constexpr (c % 2 == 0) p[c] = cmul(p[c / 2], p[c / 2]);
        else p[c] = cmul(p[c - 1], w);
    });
}

}  // namespace fc
