// k_rows_shapes.h -- which shapes of the row-packed kernels (k_neighbors_rows.h) exist, which part (k_rows_part.hip, one translation
// unit and code object per part) holds each, and their launchers.
#ifndef HMK_ROWS_SHAPES_H
#define HMK_ROWS_SHAPES_H
#include "k_neighbors_rows.h"

namespace hmk {

// -----------------------------------------------------------------------------
// shapes, parts and launchers
// -----------------------------------------------------------------------------
template <int X, int D, int CAP, bool EXACT_LB>
static hipError_t launch_rows_t(const NeighborParams &P, uint32_t tile_base, uint32_t n_tiles, hipStream_t s) {
    constexpr int G = rows_groups(X, D, CAP, EXACT_LB);
    // the flush's mode is a template parameter: with the run-time form the counting branch's registers spill in every mode
    if constexpr (!EXACT_LB && (X == 2 || X == 3)) {   // mixed lengths at the max shifts of sets with mean length 6 .. 13.9: a one-length form per tile (k_neighbors_rows_lens)
        static_assert(G == 2, "the planner gives the capacity groups 16 rows per tile");
        if (P.deg) hipLaunchKernelGGL((k_neighbors_rows_lens<X, D, CAP, EDGES_COUNT>), dim3(n_tiles), dim3(256), 0, s, P, tile_base);
        else hipLaunchKernelGGL((k_neighbors_rows_lens<X, D, CAP, EDGES_PLAIN>), dim3(n_tiles), dim3(256), 0, s, P, tile_base);
        return hipGetLastError();
    } else {
        if (P.deg)
            hipLaunchKernelGGL((k_neighbors_rows<X, D, CAP, EXACT_LB, G, EDGES_COUNT>), dim3(n_tiles), dim3(256), 0, s, P, tile_base);
        else
            hipLaunchKernelGGL((k_neighbors_rows<X, D, CAP, EXACT_LB, G, EDGES_PLAIN>), dim3(n_tiles), dim3(256), 0, s, P, tile_base);
        return hipGetLastError();
    }
}

// The instantiations.  F(part, X, L): a set of ONE length L with max shift X (column length at compile time) -- every length
// 6 .. 20 at the max shift the reference derives for it, round(L / 4) (Hammock.java:1421-1434).  C(part, X, D, CAP): the
// capacity form, column length <= CAP at run time, rows D longer (mixed lengths; also what a uniform set with another -x
// runs: D = 0).  `part` is the translation unit that holds the shape (k_rows_part.hip is compiled once per part, each a code
// object of its own: a pass loads only the parts it launches from).
#define HMK_ROWS_EXACT_LIST(F) \
    F(0, 3, 10) F(0, 3, 11) F(0, 3, 12) F(0, 3, 13) \
    F(3, 2, 6) F(3, 2, 7) F(3, 2, 8) F(3, 2, 9) \
    F(4, 4, 14) F(4, 4, 15) F(4, 4, 16) F(4, 4, 17) \
    F(6, 5, 18) F(6, 5, 19) F(6, 5, 20)
#define HMK_ROWS_CAP_LIST(C) \
    C(1, 3, 0, 12) C(1, 3, 1, 12) C(1, 3, 2, 12) C(7, 3, 3, 12) C(7, 3, 4, 12) C(7, 3, 5, 12) C(8, 3, 6, 12) C(8, 3, 7, 12) \
    C(8, 3, 8, 12) C(9, 3, 9, 12) C(9, 3, 10, 12) C(9, 3, 11, 12) C(9, 3, 12, 12) C(9, 3, 13, 12) \
    C(2, 3, 0, 16) C(2, 3, 1, 16) C(2, 3, 2, 16) C(2, 3, 3, 16) C(10, 3, 4, 16) C(10, 3, 5, 16) C(10, 3, 6, 16) C(10, 3, 7, 16) \
    C(10, 3, 0, 20) C(10, 3, 1, 20) C(10, 3, 2, 20) C(10, 3, 3, 20) \
    C(3, 1, 0, 12) C(3, 1, 1, 12) C(3, 1, 2, 12) C(3, 1, 3, 12) C(3, 1, 4, 12) \
    C(11, 2, 0, 12) C(11, 2, 1, 12) C(11, 2, 2, 12) C(11, 2, 3, 12) C(12, 2, 4, 12) C(12, 2, 5, 12) C(12, 2, 6, 12) C(12, 2, 7, 12) \
    C(3, 2, 0, 16) C(3, 2, 1, 16) C(3, 2, 2, 16) C(3, 2, 3, 16) \
    C(4, 4, 0, 12) C(4, 4, 1, 12) C(4, 4, 2, 12) C(4, 4, 3, 12) C(4, 4, 4, 12) C(4, 4, 5, 12) C(4, 4, 6, 12) C(4, 4, 7, 12) \
    C(4, 4, 8, 12) \
    C(5, 4, 0, 16) C(5, 4, 1, 16) C(5, 4, 2, 16) C(5, 4, 3, 16) C(5, 4, 4, 16) C(5, 4, 5, 16) C(5, 4, 6, 16) C(5, 4, 7, 16) \
    C(5, 4, 0, 20) C(5, 4, 1, 20) C(5, 4, 2, 20) C(5, 4, 3, 20) \
    C(6, 5, 0, 12) C(6, 5, 1, 12) C(6, 5, 2, 12) \
    C(6, 5, 0, 16) C(6, 5, 1, 16) C(6, 5, 2, 16) C(6, 5, 3, 16) C(6, 5, 4, 16) \
    C(6, 5, 0, 20) C(6, 5, 1, 20) C(6, 5, 2, 20) C(6, 5, 3, 20) C(6, 5, 4, 20)

// one launcher per part (k_rows_part.hip, -DHMK_ROWS_PART=p); hipErrorInvalidValue: no such shape in that part
#define HMK_ROWS_PART_DECL(p) \
    hipError_t launch_rows_part_##p(int X, int d, int cap, bool exact, const NeighborParams &P, uint32_t tile_base, uint32_t n_tiles, hipStream_t s);
HMK_ROWS_PART_DECL(0) HMK_ROWS_PART_DECL(1) HMK_ROWS_PART_DECL(2) HMK_ROWS_PART_DECL(3)
HMK_ROWS_PART_DECL(4) HMK_ROWS_PART_DECL(5) HMK_ROWS_PART_DECL(6) HMK_ROWS_PART_DECL(7)
HMK_ROWS_PART_DECL(8) HMK_ROWS_PART_DECL(9) HMK_ROWS_PART_DECL(10) HMK_ROWS_PART_DECL(11) HMK_ROWS_PART_DECL(12)
#undef HMK_ROWS_PART_DECL
hipError_t warm_rows_part_0();

}  // namespace hmk
#endif
