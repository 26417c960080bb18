// Tile kernel instantiations: Rec64 records, with sub-tile splitting.
#include "pc_knn_tile_kernel.h"

void pc_tile_launch_f64_split(int ks, int mode, unsigned nblocks, int nw, size_t lds, hipStream_t st, const ScanArgs &A) {
    launch_tile_ks<Rec64, true>(ks, mode, nblocks, nw, lds, st, A);
}
