// What the block plan (blockbin.hip) leaves behind for the kernels that read a tile's list from the block lists
// instead of the sorted lists: the forward blend (blockbin.hip) and the render backward (backward.hip).
#pragma once
#include "gsr_common.hpp"

namespace gsr {

constexpr int kBW = 8, kBH = 8;          // tiles per block: one lane per tile
constexpr int kUnit = 2048;              // block-list entries per unit
constexpr int kBatches = kUnit / kWave;  // 32: one lane per batch in the transposed masks
constexpr int kMaxBlocks = 512;          // 8 x 8-tile blocks per frame (4K: 30 x 17 = 510)

struct BlockMeta {                       // u32 words in HBM
    // [0, nbp]            list_start : entry index where the list of block b starts (nbp + 1 words)
    // [nbp+1, 2nbp+1]     unit_start : first unit of block b (nbp + 1 words; [nb] = total units)
    // then                ticket_count, ticket_emit (work queues of the two persistent kernels)
    // then                walked : per block, how many of its units the forward blend looked into (max over its tiles)
    uint32_t* w;
    int nbp;
    __host__ __device__ uint32_t* list_start() const { return w; }
    __host__ __device__ uint32_t* unit_start() const { return w + nbp + 1; }
    __host__ __device__ uint32_t* tickets() const { return w + 2 * (nbp + 1); }
    __host__ __device__ uint32_t* walked() const { return w + 2 * (nbp + 1) + 2; }
};
constexpr size_t kBlockMetaWords = 2 * (kMaxBlocks + 1) + 2 + kMaxBlocks;

// A tile (tx, ty) belongs to block b = (ty / 8) * nbx + tx / 8; its list is, unit after unit of the block's list and
// batch after batch (64 entries) of a unit, the entries whose bit is set in
//   unit_masks[(u * 16 + tx % 8) * 32 + batch] & unit_masks[(u * 16 + 8 + ty % 8) * 32 + batch]      (column & row mask)
// entry e of the block's list being Gaussian ent_idx[list_start[b] + e]; prefix[u * 64 + (ty % 8) * 8 + tx % 8] is the
// number of the tile's list entries in the block's units before u.
struct BlockFeed {
    BlockMeta meta;
    int nbx;
    const uint2* unit_masks;
    const uint32_t* prefix;
    const uint32_t* ent_idx;
    // scratch for per-entry gradient sums (render backward): the 8 R bytes of keysUnsorted, dead once the forward call is complete
    float* acc;
    unsigned long long acc_floats;
};
BlockFeed block_feed(int n, int grid_x, int grid_y, uint32_t r_total, char* geo_scratch, const uint32_t* ent_idx, char* bin_scratch);

// Which lists the forward call that issued `r` left for a backward call with these sizes / rows / point_list (api.hip):
// *lists_written — the sorted lists are in point_list; *from_blocks — it ran the block plan and blended from the block
// lists: *feed says where they are. GSR_ERR_INVALID_ARG if the receipt does not fit the arguments or promises no list.
// The order the gsr_forward call that issued `r` gave its blend workgroups (slow tiles first: TileOrder, blend_core.hpp), if
// it made one for exactly these rows and its tile history still holds it (the receipt's own history, or one of the calling
// thread's); else null. A hint: the render backward's tiles take long where the forward's did.
const uint32_t* tile_order_of_call(const gsr_forward_receipt& r, int row_begin, int row_end);
int lists_of_receipt(const gsr_forward_receipt& r, int n, int width, int height, int row_begin, int row_end,
                     const void* point_list, BlockFeed* feed, bool* from_blocks, bool* lists_written);

// inclusive prefix sum over lanes 0..31 (and, separately, 32..63): row_shr 1, 2, 4, 8 + row_bcast:15
__device__ __forceinline__ uint32_t prefix32_inclusive(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
    return v;
}

}  // namespace gsr
