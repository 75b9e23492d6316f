// k_neighbors_rows.hip -- which row-packed instantiation (k_neighbors_rows.h) a (max shift, lengths) class runs, and the
// dispatch to the part (k_rows_part.hip) that holds it.  Host code only.
#include "k_rows_shapes.h"

namespace hmk {

// column-length capacities of the capacity form
int rows_cap_for(int lb) { return lb <= 12 ? 12 : lb <= 16 ? 16 : lb <= 20 ? 20 : 0; }

namespace {
// part that holds the shape, or -1
int rows_part_of(int X, int d, int cap, bool exact) {
    if (exact) {
#define HMK_F(PV, XV, L) if (X == XV && d == 0 && cap == L) return PV;
        HMK_ROWS_EXACT_LIST(HMK_F)
#undef HMK_F
        return -1;
    }
#define HMK_C(PV, XV, DV, CAPV) if (X == XV && d == DV && cap == CAPV) return PV;
    HMK_ROWS_CAP_LIST(HMK_C)
#undef HMK_C
    return -1;
}
}  // namespace

int rows_per_tile_rows(int X, int d, int cap, bool exact) {
    if (rows_part_of(X, d, cap, exact) < 0) return 0;
    return 8 * rows_groups(X, d, cap, exact);
}

// is there a row-packed instantiation for this class?  exact: every sequence of the set has length lb
bool rows_kernel_available(int X, int la, int lb, bool exact) {
    if (la < lb || lb < 2 * X || X < 1) return false;
    return rows_part_of(X, la - lb, exact ? lb : rows_cap_for(lb), exact) >= 0;
}

hipError_t launch_neighbors_rows(int X, int d, int cap, bool exact, const NeighborParams &P, uint32_t tile_base,
                                 uint32_t n_tiles, hipStream_t s) {
    if (n_tiles == 0) return hipSuccess;
    switch (rows_part_of(X, d, cap, exact)) {
#define HMK_P(p) case p: return launch_rows_part_##p(X, d, cap, exact, P, tile_base, n_tiles, s);
        HMK_P(0) HMK_P(1) HMK_P(2) HMK_P(3) HMK_P(4) HMK_P(5) HMK_P(6) HMK_P(7) HMK_P(8) HMK_P(9) HMK_P(10) HMK_P(11) HMK_P(12)
#undef HMK_P
    }
    return hipErrorInvalidValue;
}

hipError_t warm_neighbors_rows_module() { return warm_rows_part_0(); }

}  // namespace hmk
