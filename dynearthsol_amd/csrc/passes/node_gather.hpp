// passes/node_gather.hpp -- LDS-staged segmented reduction used by the node passes.
// Part of the single translation unit des_dev.hip (included inside namespace des_hip, after
// DevClock / struct des_dev); not a stand-alone header.

// ---- node gathers: LDS-staged segmented reduction ---------------------------------
// A workgroup owns DES_BLOCK consecutive nodes, i.e. one contiguous range [kb, ke) of the
// CSR incidence list.  Phase 1: all lanes walk that range with stride DES_BLOCK -- the index
// loads are fully coalesced and every lane has the same number of independent record
// gathers in flight, whatever the valence of "its" node (8 or 32 on the regular mesh, 8-50
// on TetGen meshes) -- and park the gathered values in LDS.  Phase 2: each lane sums the
// slice of LDS that belongs to its node, sequentially and in ascending element order, which
// is the reference's summation order (fields.cxx:667-675) -> bit-identical sums.
// LDS slot of incidence j is skewed by j/8 so that row starts that are multiples of 8
// doubles apart (regular mesh) do not land on the same banks.
#ifndef DES_TILE_N1
#define DES_TILE_N1 768
#endif
#ifndef DES_TILE_N1C
#define DES_TILE_N1C 1024
#endif
#ifndef DES_TILE_N3
#define DES_TILE_N3 1024
#endif
#ifndef DES_TILE_N2
#define DES_TILE_N2 2048
#endif
// DES_PIPE: the record gathers of incidence tile t+1 are issued before the sums of tile t are
// formed from LDS (registers hold them across the sum), so HBM/L2 latency overlaps the LDS phase
// instead of alternating with it.
#ifndef DES_PIPE
#define DES_PIPE 1
#endif
__device__ __forceinline__ int lds_slot(int j) { return j + (j >> 3); }
#define DES_TILE_LDS(T) ((T) + (T) / 8 + 1)
