// Internal interface of the onesweep radix sort (radix_sort.hip).
#pragma once
#include <algorithm>

#include "gsr_common.hpp"

namespace gsr {

enum { kDigitBits = 0, kDigitTileX = 1, kDigitTileY = 2 };

// How a pass takes its digit out of a key.
//   kDigitBits : (key >> shift) & (nbins - 1), nbins a power of two
//   kDigitTileX: (key >> 32) % grid_x      kDigitTileY: (key >> 32) / grid_x
// (inv_grid_x = 1.0f / grid_x; exact for grids up to 255 x 255, see digit_of)
struct DigitSpec {
    int mode;
    int shift;
    uint32_t nbins;      // <= 256
    uint32_t grid_x;
    float inv_grid_x;
    uint32_t single_ticket = 0;   // set by the launcher: tiles are handed out by ONE ticket (a device too small for the XCD runs, take_tile)
};

struct SweepScratch {
    unsigned long long* status;   // tiles x radix look-back words
    uint32_t* ticket;
    uint32_t* error_word;         // set to error_value if a bounded spin ever gave up
    uint32_t error_value = 1u;    // (gsr_forward: the call's serial, so that a late writer of an OLDER call cannot raise the flag of the slot's new owner)
    uint32_t* hist;               // 8 x 256 digit counts
};

size_t sweep_scratch_bytes(size_t n);
SweepScratch carve_sweep_scratch(char* base, size_t n);

// One stable digit pass. digit_hist: nbins raw counts of this pass's digit over all n keys.
// vals_in == nullptr means "value = input index".
int sweep_pass_u64(const uint64_t* keys_in, const uint32_t* vals_in, uint64_t* keys_out, uint32_t* vals_out, uint32_t n,
                   const DigitSpec& spec, const uint32_t* digit_hist, const SweepScratch& sc, hipStream_t stream,
                   bool already_cleared = false);
int sweep_pass_u32(const uint32_t* keys_in, const uint32_t* vals_in, uint32_t* keys_out, uint32_t* vals_out, uint32_t n,
                   const DigitSpec& spec, const uint32_t* digit_hist, const SweepScratch& sc, hipStream_t stream,
                   bool already_cleared = false, const uint32_t* n_dev = nullptr, const uint32_t* vals2_in = nullptr,
                   uint32_t* vals2_out = nullptr);      // vals2_*: a second 32-bit value per key (256 bins only)
int sweep_clear(const SweepScratch& sc, uint32_t n, uint32_t nbins, hipStream_t stream);

// Digit counts of `passes` consecutive 8-bit fields starting at begin_bit (the last one
// narrowed so that no bit at or above end_bit takes part), from one read of the keys.
int histogram_bits_u32(const uint32_t* keys, size_t n, int begin_bit, int end_bit, uint32_t* hist, hipStream_t stream);
int histogram_bits_u64(const uint64_t* keys, size_t n, int begin_bit, int end_bit, uint32_t* hist, hipStream_t stream);

}  // namespace gsr
