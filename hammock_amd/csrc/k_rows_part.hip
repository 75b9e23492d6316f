// k_rows_part.hip -- one part of the row-packed kernels' instantiations (k_neighbors_rows.h), compiled once per part with
// -DHMK_ROWS_PART=p: every part is a code object of its own, loaded at the first launch from it, and the parts compile in
// parallel.  Integer scoring only: no MFMA, no dense contraction.
#include "k_rows_shapes.h"

#ifndef HMK_ROWS_PART
#error "compile with -DHMK_ROWS_PART=<part>"
#endif

namespace hmk {

namespace {
// only the shapes of THIS part are instantiated
template <bool MINE, int X, int D, int CAP, bool EXACT_LB>
struct RowsLaunch {
    static hipError_t go(const NeighborParams &, uint32_t, uint32_t, hipStream_t) { return hipErrorInvalidValue; }
};
template <int X, int D, int CAP, bool EXACT_LB>
struct RowsLaunch<true, X, D, CAP, EXACT_LB> {
    static hipError_t go(const NeighborParams &P, uint32_t tile_base, uint32_t n_tiles, hipStream_t s) {
        return launch_rows_t<X, D, CAP, EXACT_LB>(P, tile_base, n_tiles, s);
    }
};
}  // namespace

#define HMK_CAT2(a, b) a##b
#define HMK_CAT(a, b) HMK_CAT2(a, b)

hipError_t HMK_CAT(launch_rows_part_, HMK_ROWS_PART)(int X, int d, int cap, bool exact, const NeighborParams &P, uint32_t tile_base,
                                                     uint32_t n_tiles, hipStream_t s) {
    if (exact) {
#define HMK_F(PV, XV, L) \
    if (PV == HMK_ROWS_PART && X == XV && d == 0 && cap == L) return RowsLaunch<PV == HMK_ROWS_PART, XV, 0, L, true>::go(P, tile_base, n_tiles, s);
        HMK_ROWS_EXACT_LIST(HMK_F)
#undef HMK_F
        return hipErrorInvalidValue;
    }
#define HMK_C(PV, XV, DV, CAPV) \
    if (PV == HMK_ROWS_PART && X == XV && d == DV && cap == CAPV) return RowsLaunch<PV == HMK_ROWS_PART, XV, DV, CAPV, false>::go(P, tile_base, n_tiles, s);
    HMK_ROWS_CAP_LIST(HMK_C)
#undef HMK_C
    return hipErrorInvalidValue;
}

#if HMK_ROWS_PART == 0
// forces the load of the code object that holds the BASELINE shape (hmk_create)
hipError_t warm_rows_part_0() {
    hipFuncAttributes a;
    return hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&k_neighbors_rows<3, 0, 12, true, rows_groups(3, 0, 12, true), EDGES_PLAIN>));
}
#endif

}  // namespace hmk
