// Winograd F(2x2, 3x3) with a FLAT workgroup tile: one tile row of 64 tiles = 2 x 128 output pixels (round 6).
//
// The row split of PCONV.tile_conv2d sends the whole 8-row blocks of a 66 / 34 / 18 / 10-row layer to F(4x2, 3x3)
// (wino42.hip) and the 2-row remainder to F(2x2, 3x3) -- whose 4 x 64 workgroup tile (wino.hip) ran those two rows
// half empty.  This is the same kernel compiled with TX = 64, TY = 1: same arithmetic per output (the transforms, the
// sixteen GEMMs, the exchange rounds and the epilogue are shared source), a 4 x 130 patch per channel (five DMA
// pieces per thread and chunk instead of four), accumulator half n = tile columns 32 n .. 32 n + 31 instead of tile
// row n.  Exports pconv_conv3x3_wino_flat only; weights are packed by wino.hip's pconv_wino_pack_weight.
#define PCONV_WINO_FLAT 1
#include "wino.hip"
