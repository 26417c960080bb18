// Tile kernel instantiations: Rec32 records, with sub-tile splitting.
#include "pc_knn_tile_kernel.h"

void pc_tile_launch_f32_split(int ks, int mode, unsigned nblocks, int nw, size_t lds, hipStream_t st, const ScanArgs &A) {
    launch_tile_ks<Rec32, true>(ks, mode, nblocks, nw, lds, st, A);
}
