// Onesweep LSD radix sort, LDS-staged, written for gfx950 (no CUB / hipCUB / rocPRIM).
// Replaces cub::DeviceRadixSort::SortPairs at reference GSCuda.cu:794-797.
//
// One kernel launch per digit pass moves every key exactly once (12 B read + 12 B written
// per u64/u32 pair): a workgroup takes the next 4096-key tile from an atomic ticket, ranks
// its keys in LDS, publishes its per-digit counts and resolves its global offsets by
// decoupled look-back over the tiles before it, then writes each digit's run as one
// contiguous burst. Digit histograms of all passes come from one up-front read of the keys
// (or, for the tile passes of the frame pipeline, analytically from the Gaussians'
// rectangles — binning.hip — with no key traffic at all).
//
// Ranking: wave64 __ballot match groups + one LDS counter per (wave, digit); a tile's keys
// are ranked in (wave, item, lane) order = index order, and tiles are ordered by ticket, so
// the sort is stable (equal keys keep ascending input order).
//
// Cross-workgroup protocol: only the 64-bit status words travel between workgroups, each
// written by ONE relaxed agent-scope atomic store and read by relaxed agent-scope atomic
// loads (data-is-the-flag granules, cdna_hip_programming.md Guideline 16 R2); no payload is
// exchanged inside the launch, so no fences are needed. Tickets make predecessors resident
// before their successors; every spin is bounded and raises an error word instead of hanging.
#include "gsr_common.hpp"
#include "radix_sort.hpp"

namespace gsr {
int sweep_clear(const SweepScratch& sc, uint32_t n, uint32_t nbins, hipStream_t stream);
namespace {

constexpr int kThreads = 512;
constexpr int kWaves = kThreads / kWave;
#ifndef GSR_SORT_ITEMS
#define GSR_SORT_ITEMS 16
#endif
#ifndef GSR_SORT_WAVES
#define GSR_SORT_WAVES 4                                // waves per SIMD the pass kernels are compiled for (two workgroups per CU)
#endif
constexpr int kItems = GSR_SORT_ITEMS;                  // keys per lane (measured: 8192-key tiles beat 4096 and 2048)
constexpr int kSortTile = kThreads * kItems;            // keys per workgroup
constexpr int kWaveSpan = kWave * kItems;
// The ranked tile leaves through LDS in kStageRounds slices of kStageSlots slots: a smaller LDS
// footprint per workgroup buys more resident workgroups per CU (the pass is latency-bound).
constexpr int kStageRounds = 2;
constexpr int kStageSlots = kSortTile / kStageRounds;
static_assert(kItems % kStageRounds == 0, "items per lane must split evenly over the staging rounds");

constexpr unsigned long long kFlagAggregate = 1ull << 62;
constexpr unsigned long long kFlagPrefix = 2ull << 62;
constexpr unsigned long long kValueMask = (1ull << 62) - 1ull;
constexpr uint32_t kSpinLimit = 1u << 22;
constexpr int kLookWindow = 4;                          // predecessors' status words read per look-back round
constexpr int kLookLanes = 1;                           // lanes sharing one digit's look-back

template <typename KeyT>
__device__ __forceinline__ uint32_t digit_of(KeyT key, const DigitSpec& s) {
    if (s.mode == kDigitBits) return (uint32_t)(key >> s.shift) & (s.nbins - 1u);
    // tile / grid_x in float: (tile + 0.5) / grid_x sits at least 0.5 / 255 away from an integer, the
    // float error is below 1e-4 for tile < 2^16, so the truncation is exact (grids up to 255 x 255);
    // four full-rate operations instead of a quarter-rate 32-bit multiply-high.
    const uint32_t tile = (uint32_t)((unsigned long long)key >> 32);
    const uint32_t y = (uint32_t)(((float)tile + 0.5f) * s.inv_grid_x);
    return s.mode == kDigitTileX ? tile - __umul24(y, s.grid_x) : y;
}

// ---- histograms of all bit-field passes from one read of the keys -----------------------
template <typename KeyT>
__global__ __launch_bounds__(kThreads) void histogram_bits_kernel(const KeyT* __restrict__ keys, size_t n, int passes,
                                                                   int first_shift, int end_bit, uint32_t* __restrict__ hist) {
    __shared__ uint32_t lds[8 * 256];
    for (int i = threadIdx.x; i < passes * 256; i += kThreads) lds[i] = 0;
    __syncthreads();
    const size_t stride = (size_t)gridDim.x * kThreads;
    const size_t rounds = (n + stride - 1) / stride;
    for (size_t it = 0; it < rounds; ++it) {
        const size_t e = it * stride + (size_t)blockIdx.x * kThreads + threadIdx.x;
        const bool live = e < n;
        const KeyT k = live ? keys[e] : (KeyT)0;
        for (int p = 0; p < passes; ++p) {
            const int shift = first_shift + 8 * p;
            const uint32_t mask = (end_bit - shift >= 8) ? 255u : ((1u << (end_bit - shift)) - 1u);
            const uint32_t d = (uint32_t)(k >> shift) & mask;
            if (p == passes - 1 && passes > 1) {
                // The top digit of real keys (float bit patterns, tile ids) takes few distinct values: one
                // LDS atomic per distinct value and wave instead of 64 colliding on the same counter.
                unsigned long long todo = __ballot(live);
                while (todo) {
                    const uint32_t v = (uint32_t)__shfl((int)d, __ffsll((long long)todo) - 1, kWave);
                    const unsigned long long same = __ballot(live && d == v) & todo;
                    if ((threadIdx.x & (kWave - 1)) == (uint32_t)(__ffsll((long long)same) - 1))
                        atomicAdd(&lds[p * 256 + v], (uint32_t)__popcll(same));
                    todo &= ~same;
                }
            } else if (live) {
                atomicAdd(&lds[p * 256 + d], 1u);
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < passes * 256; i += kThreads)
        if (lds[i]) atomicAdd(&hist[i], lds[i]);
}

// ---- one digit pass ----------------------------------------------------------------------
// Look-back state of one digit group, resumable so that its round trips can be interleaved
// with the ranking and staging work of the tile.
template <int RADIX, int LPD>
struct LookBack {
    unsigned long long sw[kLookWindow];
    unsigned long long excl = 0;
    uint32_t t = 0;          // next predecessor to consume is t - 1
    uint32_t spins = 0;
    bool active = false;     // this lane takes part
    bool found = false;
    int d = 0, sub = 0;

    __device__ __forceinline__ void issue(const unsigned long long* status) {
#pragma unroll
        for (int k = 0; k < kLookWindow; ++k) {
            const uint32_t back = (uint32_t)(sub * kLookWindow + k);
            const uint32_t tk = (t > back) ? t - 1 - back : 0u;
            sw[k] = __hip_atomic_load(status + (size_t)tk * RADIX + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // Consumes the words of the round in flight, in tile order. Returns the tiles consumed.
    __device__ __forceinline__ uint32_t consume(int lane) {
        unsigned long long seg_sum = 0;
        uint32_t seg_used = 0;
        bool seg_found = false;
#pragma unroll
        for (int k = 0; k < kLookWindow; ++k) {
            const uint32_t back = (uint32_t)(sub * kLookWindow + k);
            if (!seg_found && seg_used == (uint32_t)k && t > back) {
                const unsigned long long f = sw[k] & ~kValueMask;
                if (f != 0) {
                    seg_sum += sw[k] & kValueMask;
                    seg_used = (uint32_t)k + 1;
                    if (f == kFlagPrefix) seg_found = true;
                }
            }
        }
        uint32_t consumed = 0;
        if (LPD == 1) {
            excl += seg_sum;
            consumed = seg_used;
            found = seg_found;
        } else {
            const int group_lane0 = lane - sub;
            bool open = true;
#pragma unroll
            for (int j = 0; j < LPD; ++j) {
                const unsigned long long js = __shfl(seg_sum, group_lane0 + j, kWave);
                const uint32_t ju = __shfl(seg_used, group_lane0 + j, kWave);
                const int jf = __shfl((int)seg_found, group_lane0 + j, kWave);
                if (open) {
                    excl += js;
                    consumed += ju;
                    if (jf) { found = true; open = false; }
                    else if (ju != (uint32_t)kLookWindow) open = false;
                }
            }
        }
        t -= min(consumed, t);                   // tile 0 always carries kFlagPrefix
        return consumed;
    }
};

// Which tile a workgroup takes. Tiles that follow each other write adjacent runs of every digit's region, and a run
// (32 keys on average) ends inside a cache line: with tile = ticket the two halves of such a line are written by
// workgroups on different XCDs, each L2 sends its half to memory as a partial line, and the pass is bound by those
// (scripts/micro/scatter_records.hip, 46 M records: the write side alone 0.243 ms with neighbouring tiles on different
// XCDs, 0.173 ms with eight consecutive tiles on one). Workgroups b and b + 8 share an XCD (observed round-robin
// placement, MI355X_MICROARCH.md: speed only, never correctness), so each residue class r = blockIdx % 8 keeps a ticket
// of its own and the j-th workgroup of class r to start takes position 8 j + r of a sequence in which every run of
// kXcdRun consecutive TILES belongs to one class. Whatever the placement, every position below `tiles` is taken exactly
// once (a class has as many workgroups as positions, the grid being at least `tiles` wide). No deadlock, GIVEN that the
// device starts workgroups in the order of their numbers and has room for more than one group of them: a tile waits only
// for tiles below it; the lowest unfinished tile, if nobody has taken it yet, is the next position of its class, and the
// workgroups that can be waiting for it are those of ITS group of 8 kXcdRun tiles (earlier groups hold lower tiles, all
// finished) — at most 7 kXcdRun of the chip's 512 slots — so the launch has room to start the next workgroups, one of
// which is of that class. A device that cannot hold two such groups (fewer than 64 compute units at two workgroups each:
// a partition, a CU mask) gets the single ticket instead (`single`: tile = ticket, progress under ANY placement — a
// workgroup that holds a ticket is running, and the lowest ticket never waits); the launcher decides from the CU count.
// Every spin is bounded either way: a placement nobody foresaw raises GSR_ERR_INTERNAL instead of hanging.
#ifndef GSR_XCD_RUN
#define GSR_XCD_RUN 8
#endif
constexpr uint32_t kXcdRun = GSR_XCD_RUN;
__device__ __forceinline__ uint32_t take_tile(uint32_t* ticket, uint32_t tiles, bool single) {
    if (single) return atomicAdd(ticket, 1u);
    const uint32_t r = blockIdx.x & 7u;
    const uint32_t j = atomicAdd(ticket + r, 1u);
    const uint32_t pos = 8u * j + r;
    const uint32_t full = tiles / (8u * kXcdRun) * (8u * kXcdRun);
    // (positions in the last, incomplete group — and beyond the last tile: those workgroups leave at once — map to themselves)
    return pos < full ? (j / kXcdRun) * (8u * kXcdRun) + r * kXcdRun + (j % kXcdRun) : pos;
}

// SECOND: every key carries a second 32-bit value (vals2_in -> vals2_out; the depth order's packed rectangle), staged and
// written beside the first. It is loaded after the ranking loop: 16 more registers across that loop would leave one
// workgroup per CU instead of two.
// DROP (the depth order's first pass on scenes beyond 16 M Gaussians, which then needs no compaction in front of it): the
// input is the preprocess's per-Gaussian array itself — keys of Gaussians without a tile are the 0xFFFFFFFF sentinel and take
// no part (no rank, no slot: the tile's live keys leave in index order as if the others were not there), a key's value is its
// position, and the few keys with another top byte than the main one go to the side list (DepthSide) as the compaction
// would have sent them. drop.n_out: the device word with the number of keys that do take part (bounds the stores).
struct DropSpec {
    const uint32_t* n_out = nullptr;
    DepthSide side;
};
// REC (u32 keys with a second value only; GSR_DEPTH_RECORDS): the triples travel between the passes as 12-byte RECORDS
// {key, value, second value} instead of three arrays — bit 0 (kRecIn): keys_in is the record array (vals_in, vals2_in unused);
// bit 1 (kRecOut): keys_out is. A digit's run of a tile (32 keys on average) then leaves as ONE piece of 384 bytes instead of
// three of 128, each of which ends inside a cache line (`profiles/r06_micro_scatter_records.txt`: 5-6 % of a pass of 46 M).
enum { kRecIn = 1, kRecOut = 2 };
struct __attribute__((aligned(4))) DepthRecord { uint32_t key, val, second; };
template <typename KeyT, int BITS, bool SECOND, bool DROP = false, int REC = 0>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(GSR_SORT_WAVES))) void onesweep_kernel(const KeyT* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                                                            KeyT* __restrict__ keys_out, uint32_t* __restrict__ vals_out,
                                                            const uint32_t* __restrict__ vals2_in, uint32_t* __restrict__ vals2_out,
                                                            uint32_t n_host, const uint32_t* __restrict__ n_dev, const DigitSpec spec,
                                                            const uint32_t* __restrict__ digit_hist,
                                                            unsigned long long* status, uint32_t* ticket,
                                                            uint32_t* error_word, uint32_t error_value, const DropSpec drop = DropSpec()) {
    // n_dev (may be null): the key count lives on the device — the pass was queued before the host knew it, with a
    // grid sized for an upper bound; workgroups whose ticket lies beyond the last tile leave at once.
    const uint32_t n = n_dev ? *n_dev : n_host;
    constexpr int RADIX = 1 << BITS;
    constexpr int kLanesPerDigit = (kThreads / RADIX) < kLookLanes ? (kThreads / RADIX) : kLookLanes;
    __shared__ uint32_t wave_hist[kWaves][RADIX];
    __shared__ uint32_t tile_hist[RADIX];        // digit counts of the tile (early, by LDS atomics)
    __shared__ uint32_t run_start[RADIX];        // first slot of digit d inside the ranked tile
    __shared__ uint32_t global_start[RADIX];     // output index of this tile's first digit-d key
    __shared__ uint32_t run_delta[RADIX];        // global_start - run_start (u32 wrap-around)
    __shared__ uint32_t scan_ws[kWaves];
    __shared__ uint32_t s_tile, s_fail, s_live;
    __shared__ KeyT stage_keys[kStageSlots];
    __shared__ uint32_t stage_vals[kStageSlots];
    __shared__ uint32_t stage_vals2[SECOND ? kStageSlots : 1];

    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    if (threadIdx.x == 0) {
        s_tile = take_tile(ticket, (n + (uint32_t)kSortTile - 1u) / (uint32_t)kSortTile, spec.single_ticket != 0u);
        s_fail = 0;
    }
    if (threadIdx.x < RADIX) {
#pragma unroll
        for (int w = 0; w < kWaves; ++w) wave_hist[w][threadIdx.x] = 0;
        tile_hist[threadIdx.x] = 0;
    }
    __syncthreads();
    const uint32_t tile = s_tile;
    const uint32_t tile_base = tile * (uint32_t)kSortTile;
    if (tile_base >= n) return;                  // cannot happen with grid = ceil(n / tile)
    const uint32_t valid = min((uint32_t)kSortTile, n - tile_base);

    // Issue every global load of the tile first (keys AND values): the HBM latency is paid once.
    KeyT key[kItems];
    uint32_t val[kItems];
    uint32_t rd[kItems];                         // rank (low 16 bits) | digit (high 16 bits)
    uint32_t val2[SECOND ? kItems : 1];
    if constexpr (REC & kRecIn) {
        // (a record per lane and load: the wave reads 768 consecutive bytes)
#pragma unroll
        for (int i = 0; i < kItems; ++i) {
            const uint32_t local = (uint32_t)(wave * kWaveSpan + i * kWave + lane);
            DepthRecord r = {0u, tile_base + local, 0u};
            if (local < valid) r = reinterpret_cast<const DepthRecord*>(keys_in)[tile_base + local];
            key[i] = (KeyT)r.key; val[i] = r.val; val2[SECOND ? i : 0] = r.second;
        }
    } else {
#pragma unroll
        for (int i = 0; i < kItems; ++i) {
            const uint32_t local = (uint32_t)(wave * kWaveSpan + i * kWave + lane);
            key[i] = (local < valid) ? keys_in[tile_base + local] : (KeyT)0;
        }
#pragma unroll
        for (int i = 0; i < kItems; ++i) {
            const uint32_t local = (uint32_t)(wave * kWaveSpan + i * kWave + lane);
            val[i] = (local < valid && vals_in) ? vals_in[tile_base + local] : tile_base + local;
        }
    }
    const uint32_t hist_c = (threadIdx.x < spec.nbins) ? digit_hist[threadIdx.x] : 0u;
    uint32_t live_bits = 0;                      // DROP: bit i = item i of this lane takes part
    if constexpr (DROP) {
        const bool side_on = drop.side.words != nullptr && drop.side.words[0] != 0u;
#pragma unroll
        for (int i = 0; i < kItems; ++i) {
            const uint32_t local = (uint32_t)(wave * kWaveSpan + i * kWave + lane);
            bool live = local < valid && (uint32_t)key[i] != 0xFFFFFFFFu;
            if (live && side_on && ((uint32_t)key[i] >> 24) != drop.side.main_top) {
                // (a handful per frame; any order: depth_side_kernel ranks them by key and index)
                const uint32_t slot = atomicAdd(&drop.side.words[1], 1u);
                if (slot < drop.side.capacity) {
                    drop.side.keys[slot] = (uint32_t)key[i];
                    drop.side.vals[slot] = tile_base + local;
                    if (drop.side.rects) drop.side.rects[slot] = vals2_in ? vals2_in[tile_base + local] : 0u;
                }
                live = false;                    // (how many of them lie below the main top byte: counted by depth_hist_kernel, for the host)
            }
            live_bits |= live ? (1u << i) : 0u;
        }
    }

    // exclusive scan of the global digit histogram -> first output index of every digit
    {
        uint32_t incl = hist_c;
#pragma unroll
        for (int off = 1; off < kWave; off <<= 1) {
            const uint32_t o = __shfl_up(incl, off, kWave);
            if (lane >= off) incl += o;
        }
        if (lane == kWave - 1) scan_ws[wave] = incl;
        __syncthreads();
        uint32_t wbase = 0;
#pragma unroll
        for (int w = 0; w < kWaves; ++w)
            if (w < wave) wbase += scan_ws[w];
        if (threadIdx.x < RADIX) global_start[threadIdx.x] = wbase + incl - hist_c;
        __syncthreads();
    }

    // Stable ranks: wave64 match groups + per-wave LDS counters. Per digit bit one ballot and, per
    // 32-lane half, one xor + or: a lane keeps the lanes whose bit equals its own. Every lane of a
    // group reads the group's counter, the lowest lane then adds the group size (LDS operations
    // of one wave execute in order, so the reads see the value before the update).
#pragma unroll
    for (int i = 0; i < kItems; ++i) {
        const uint32_t local = (uint32_t)(wave * kWaveSpan + i * kWave + lane);
        const bool live = DROP ? ((live_bits >> i) & 1u) != 0u : true;
        // padding of the last tile ranks after every real key: top digit, highest indices (DROP: a key that takes no part
        // has no rank at all — it is taken out of every lane's peers)
        const uint32_t d = DROP ? (live ? digit_of<KeyT>(key[i], spec) : 0u)
                                : ((local < valid) ? digit_of<KeyT>(key[i], spec) : (uint32_t)(RADIX - 1));
        uint32_t peers_lo = ~0u, peers_hi = ~0u;
#pragma unroll
        for (int b = 0; b < BITS; ++b) {
            const int m = __builtin_amdgcn_sbfe((int)d, b, 1);                 // 0 or -1
            const unsigned long long bal = __ballot(m != 0);
            peers_lo &= ~((uint32_t)bal ^ (uint32_t)m);
            peers_hi &= ~((uint32_t)(bal >> 32) ^ (uint32_t)m);
        }
        if constexpr (DROP) {
            const unsigned long long lm = __ballot(live);
            peers_lo &= (uint32_t)lm;
            peers_hi &= (uint32_t)(lm >> 32);
        }
        const uint32_t below = __builtin_amdgcn_mbcnt_hi(peers_hi, __builtin_amdgcn_mbcnt_lo(peers_lo, 0u));
        if (live) {
            const uint32_t prior = wave_hist[wave][d];
            if (below == 0) wave_hist[wave][d] = prior + (uint32_t)__popc(peers_lo) + (uint32_t)__popc(peers_hi);
            rd[i] = (d << 16) | (prior + below);
        } else {
            rd[i] = 0xFFFFu;                     // (no slot: never staged)
        }
    }
    __syncthreads();
    // The second values come in now, with the ranking loop behind (see SECOND above): their round trip is hidden
    // by the digit scan and the look-back.
    if constexpr (SECOND && !(REC & kRecIn)) {
        // (the base is opaque to the compiler: the loads are issued HERE, not hoisted to the top of the kernel)
        uint32_t first = tile_base + (uint32_t)(wave * kWaveSpan + lane);
        asm volatile("" : "+v"(first));
#pragma unroll
        for (int i = 0; i < kItems; ++i) {
            const uint32_t e = first + (uint32_t)(i * kWave);
            val2[i] = (e < tile_base + valid) ? vals2_in[e] : 0u;
        }
    }

    // per digit: exclusive offsets across waves (-> the tile's count), publish it, start the
    // look-back, then the exclusive scan across digits (padding sits in the top digit)
    LookBack<RADIX, kLanesPerDigit> lb;
    lb.d = threadIdx.x / kLanesPerDigit;
    lb.sub = threadIdx.x % kLanesPerDigit;
    lb.t = tile;
    lb.active = tile != 0 && threadIdx.x < RADIX * kLanesPerDigit && (uint32_t)lb.d < spec.nbins;
    {
        uint32_t acc = 0;
        if (threadIdx.x < RADIX) {
#pragma unroll
            for (int w = 0; w < kWaves; ++w) {
                const uint32_t c = wave_hist[w][threadIdx.x];
                wave_hist[w][threadIdx.x] = acc;
                acc += c;
            }
            // Padding keys were counted in the top digit: they are not published.
            const uint32_t real = (!DROP && threadIdx.x == RADIX - 1) ? acc - ((uint32_t)kSortTile - valid) : acc;
            tile_hist[threadIdx.x] = real;
            if (threadIdx.x < spec.nbins)
                __hip_atomic_store(status + (size_t)tile * RADIX + threadIdx.x,
                                   (tile == 0 ? kFlagPrefix : kFlagAggregate) | real, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        }
        uint32_t incl = acc;
#pragma unroll
        for (int off = 1; off < kWave; off <<= 1) {
            const uint32_t o = __shfl_up(incl, off, kWave);
            if (lane >= off) incl += o;
        }
        if (lane == kWave - 1) scan_ws[wave] = incl;
        __syncthreads();
        if (lb.active) lb.issue(status);         // round 1 flies during the slot computation
        uint32_t wbase = 0;
#pragma unroll
        for (int w = 0; w < kWaves; ++w)
            if (w < wave) wbase += scan_ws[w];
        if (threadIdx.x < RADIX) run_start[threadIdx.x] = wbase + incl - acc;
        if (DROP && threadIdx.x == RADIX - 1) s_live = wbase + incl;      // (the tile's keys that take part)
    }
    __syncthreads();
    const uint32_t valid_out = DROP ? s_live : valid;
    // final slot of every key inside the ranked tile (kept in the low half of rd)
#pragma unroll
    for (int i = 0; i < kItems; ++i) {
        const uint32_t d = rd[i] >> 16;
        if (!DROP || ((live_bits >> i) & 1u))
            rd[i] = (rd[i] & 0xFFFF0000u) | ((rd[i] & 0xFFFFu) + run_start[d] + wave_hist[wave][d]);
    }

    if (lb.active) {
        while (!lb.found) {
            const uint32_t used = lb.consume(lane);
            if (lb.found) break;
            if (used == 0) {
                if (++lb.spins > kSpinLimit) { s_fail = 1; *reinterpret_cast<volatile uint32_t*>(error_word) = error_value; break; }      // (may be host memory)
                __builtin_amdgcn_s_sleep(1);
            }
            lb.issue(status);
        }
        if (lb.sub == 0 && lb.found) {
            __hip_atomic_store(status + (size_t)tile * RADIX + lb.d, kFlagPrefix | (lb.excl + tile_hist[lb.d]),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            global_start[lb.d] += (uint32_t)lb.excl;
        }
    }
    __syncthreads();
    if (s_fail) return;
    const uint32_t n_out = DROP ? *drop.n_out : n;
    if (threadIdx.x < RADIX) run_delta[threadIdx.x] = global_start[threadIdx.x] - run_start[threadIdx.x];
    __syncthreads();

    // Slice r of the ranked tile: slots [r * kStageSlots, (r + 1) * kStageSlots) pass through LDS and
    // leave as contiguous runs per digit.
#pragma unroll
    for (int r = 0; r < kStageRounds; ++r) {
        if (r > 0) __syncthreads();
#pragma unroll
        for (int i = 0; i < kItems; ++i) {
            const uint32_t slot = (rd[i] & 0xFFFFu) - (uint32_t)(r * kStageSlots);
            if (slot < (uint32_t)kStageSlots) {
                stage_keys[slot] = key[i];
                stage_vals[slot] = val[i];
                if constexpr (SECOND) stage_vals2[slot] = val2[i];
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < kItems / kStageRounds; ++i) {
            const uint32_t q = (uint32_t)(i * kThreads) + threadIdx.x;          // slot inside the slice
            const uint32_t p = q + (uint32_t)(r * kStageSlots);                  // slot inside the tile
            if (p < valid_out) {
                const KeyT k = stage_keys[q];
                const uint32_t d = digit_of<KeyT>(k, spec);
                const uint32_t dst = p + run_delta[d];                       // global start - start in tile
                if (dst < n_out) {
                    if constexpr (REC & kRecOut) {
                        reinterpret_cast<DepthRecord*>(keys_out)[dst] = DepthRecord{(uint32_t)k, stage_vals[q], stage_vals2[q]};
                    } else {
                        keys_out[dst] = k;
                        vals_out[dst] = stage_vals[q];
                        if constexpr (SECOND) vals2_out[dst] = stage_vals2[q];
                    }
                }
            }
        }
    }
}

inline int radix_bits_for(uint32_t nbins) {
    int bits = 1;
    while ((1u << bits) < nbins) ++bits;
    return bits < 4 ? 4 : bits;
}

// (two workgroups of eight waves a CU; the XCD runs want room for two groups of 8 kXcdRun workgroups: take_tile)
bool xcd_runs_fit(int cus) { return 2u * (uint32_t)cus >= 2u * 8u * kXcdRun; }

template <typename KeyT>
int launch_pass(const KeyT* keys_in, const uint32_t* vals_in, KeyT* keys_out, uint32_t* vals_out, uint32_t n,
                const DigitSpec& spec_in, const uint32_t* digit_hist, const SweepScratch& sc, hipStream_t stream,
                bool already_cleared, const uint32_t* n_dev = nullptr, const uint32_t* vals2_in = nullptr,
                uint32_t* vals2_out = nullptr, const DropSpec* drop = nullptr, int rec = 0) {
    DigitSpec spec = spec_in;
    {
        DeviceShape shape;
        const int rc = current_device_shape(&shape);
        if (rc != GSR_OK) return rc;
        spec.single_ticket = xcd_runs_fit(shape.cus) ? 0u : 1u;
    }
    const int bits = radix_bits_for(spec.nbins);
    if (bits > 8) return GSR_ERR_INVALID_ARG;
    const uint32_t tiles = (n + kSortTile - 1) / kSortTile;
    if (!already_cleared) {
        const int rc = sweep_clear(sc, n, spec.nbins, stream);
        if (rc != GSR_OK) return rc;
    }
#define GSR_SWEEP(B, SECOND)                                                                                          \
    hipLaunchKernelGGL((onesweep_kernel<KeyT, B, SECOND>), dim3(tiles), dim3(kThreads), 0, stream, keys_in, vals_in, \
                       keys_out, vals_out, vals2_in, vals2_out, n, n_dev, spec, digit_hist, sc.status, sc.ticket, sc.error_word, sc.error_value)
    if (vals2_in) {
        // a second value per key: the depth passes only (u32 keys, 256 bins)
        if constexpr (sizeof(KeyT) == 4) {
            if (bits != 8 || !vals2_out) return GSR_ERR_INVALID_ARG;
#define GSR_SWEEP_REC(DROP, REC, DROPSPEC)                                                                                          \
    hipLaunchKernelGGL((onesweep_kernel<KeyT, 8, true, DROP, REC>), dim3(tiles), dim3(kThreads), 0, stream, keys_in, vals_in, keys_out, \
                       vals_out, vals2_in, vals2_out, n, n_dev, spec, digit_hist, sc.status, sc.ticket, sc.error_word, sc.error_value, DROPSPEC)
            if (drop) {
                if (vals_in || n_dev || !drop->n_out || (rec & kRecIn)) return GSR_ERR_INVALID_ARG;      // (the value is the key's position; n is the array's length)
                if (rec & kRecOut) GSR_SWEEP_REC(true, kRecOut, *drop);
                else GSR_SWEEP_REC(true, 0, *drop);
            } else if (rec == (kRecIn | kRecOut)) {
                GSR_SWEEP_REC(false, kRecIn | kRecOut, DropSpec());
            } else if (rec == kRecIn) {
                GSR_SWEEP_REC(false, kRecIn, DropSpec());
            } else if (rec == kRecOut) {
                GSR_SWEEP_REC(false, kRecOut, DropSpec());
            } else {
                GSR_SWEEP(8, true);
            }
#undef GSR_SWEEP_REC
        } else {
            return GSR_ERR_INVALID_ARG;
        }
    } else {
        if (rec) return GSR_ERR_INVALID_ARG;
        switch (bits) {
            case 4: GSR_SWEEP(4, false); break;
            case 5: GSR_SWEEP(5, false); break;
            case 6: GSR_SWEEP(6, false); break;
            case 7: GSR_SWEEP(7, false); break;
            default: GSR_SWEEP(8, false); break;
        }
    }
#undef GSR_SWEEP
    GSR_LAUNCH_CHECK("onesweep_kernel");
    return GSR_OK;
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace

size_t sweep_scratch_bytes(size_t n) {
    const size_t tiles = (n + kSortTile - 1) / kSortTile;
    return align_up(tiles * 256 * sizeof(unsigned long long), 128) + 128 /*ticket*/ + 128 /*error*/ +
           align_up(8 * 256 * sizeof(uint32_t), 128);
}

SweepScratch carve_sweep_scratch(char* base, size_t n) {
    const size_t tiles = (n + kSortTile - 1) / kSortTile;
    SweepScratch s;
    size_t off = 0;
    s.ticket = reinterpret_cast<uint32_t*>(base + off); off += 128;     // directly in front of the status
    s.status = reinterpret_cast<unsigned long long*>(base + off);      // words: one clear covers both
    off += align_up(tiles * 256 * sizeof(unsigned long long), 128);
    s.error_word = reinterpret_cast<uint32_t*>(base + off); off += 128;
    s.hist = reinterpret_cast<uint32_t*>(base + off); off += align_up(8 * 256 * sizeof(uint32_t), 128);
    return s;
}

// Zeroes the look-back words and the ticket of one pass (done inside sweep_pass_* unless the
// caller has already done it, e.g. to keep the clears out of a timed region).
int sweep_clear(const SweepScratch& sc, uint32_t n, uint32_t nbins, hipStream_t stream) {
    const uint32_t tiles = (n + kSortTile - 1) / kSortTile;
    const size_t status_bytes = (size_t)tiles * ((size_t)1 << radix_bits_for(nbins)) * sizeof(unsigned long long);
    GSR_HIP_TRY(hipMemsetAsync(sc.ticket, 0, 128 + status_bytes, stream));
    return GSR_OK;
}

int sweep_pass_u64(const uint64_t* keys_in, const uint32_t* vals_in, uint64_t* keys_out, uint32_t* vals_out, uint32_t n,
                   const DigitSpec& spec, const uint32_t* digit_hist, const SweepScratch& sc, hipStream_t stream,
                   bool already_cleared) {
    return launch_pass<unsigned long long>(reinterpret_cast<const unsigned long long*>(keys_in), vals_in,
                                           reinterpret_cast<unsigned long long*>(keys_out), vals_out, n, spec, digit_hist, sc,
                                           stream, already_cleared);
}

int sweep_pass_u32(const uint32_t* keys_in, const uint32_t* vals_in, uint32_t* keys_out, uint32_t* vals_out, uint32_t n,
                   const DigitSpec& spec, const uint32_t* digit_hist, const SweepScratch& sc, hipStream_t stream,
                   bool already_cleared, const uint32_t* n_dev, const uint32_t* vals2_in, uint32_t* vals2_out) {
    return launch_pass<uint32_t>(keys_in, vals_in, keys_out, vals_out, n, spec, digit_hist, sc, stream, already_cleared, n_dev,
                                 vals2_in, vals2_out);
}

template <typename KeyT>
static int histogram_bits(const KeyT* keys, size_t n, int first_shift, int end_bit, uint32_t* hist, hipStream_t stream) {
    const int passes = (end_bit - first_shift + 7) / 8;
    if (passes < 1 || passes > 8) return GSR_ERR_INVALID_ARG;
    GSR_HIP_TRY(hipMemsetAsync(hist, 0, (size_t)passes * 256 * sizeof(uint32_t), stream));
    const unsigned blocks = (unsigned)std::min<size_t>((n + kSortTile - 1) / kSortTile, 2048);
    hipLaunchKernelGGL((histogram_bits_kernel<KeyT>), dim3(blocks ? blocks : 1), dim3(kThreads), 0, stream, keys, n, passes,
                       first_shift, end_bit, hist);
    GSR_LAUNCH_CHECK("histogram_bits_kernel");
    return GSR_OK;
}

int histogram_bits_u32(const uint32_t* keys, size_t n, int begin_bit, int end_bit, uint32_t* hist, hipStream_t stream) {
    return histogram_bits<uint32_t>(keys, n, begin_bit, end_bit, hist, stream);
}
int histogram_bits_u64(const uint64_t* keys, size_t n, int begin_bit, int end_bit, uint32_t* hist, hipStream_t stream) {
    return histogram_bits<unsigned long long>(reinterpret_cast<const unsigned long long*>(keys), n, begin_bit, end_bit, hist, stream);
}

// ---- depth order: stable sort of N u32 keys carrying their own index ---------------------
// in -> a -> b -> a -> b : the result (sorted keys, original indices) is in (b_k, b_v).
namespace {
// How many distinct digits the top byte of the visible keys takes (one thread per digit; culled Gaussians'
// 0xFFFFFFFF sentinels are compacted away before the histogram, so digit 255 is a real depth — the top byte
// of a negative NaN, which passes the reference's frustum test — and counts like any other).
// (out[4]: the frame's count of tiles with a list, accumulated later by the tile-range kernel, starts at zero here)
// (host_top, may be null: mapped host memory that gets the count too — read by the host after an event, no copy command)
// (side, may be null: the side list's words — [1] keys on it, [2] those of them below the main top byte; host_top[8] gets [2])
__global__ __launch_bounds__(256) void top_digit_count_kernel(const uint32_t* __restrict__ hist_top, uint32_t* __restrict__ out,
                                                              uint32_t* __restrict__ host_top, const uint32_t* __restrict__ side) {
    const int c = __syncthreads_count(hist_top[threadIdx.x] != 0u);
    if (threadIdx.x == 0) {
        out[0] = (uint32_t)c;
        out[4] = 0u;
        if (host_top) {
            host_top[0] = (uint32_t)c;
            if (side) { host_top[8] = side[2]; host_top[10] = side[1]; }     // (keys below the main top byte; keys the compaction PUT on the side list)
        }
    }
}

// One LDS atomic per distinct value and wave instead of one per lane: for digits that take FEW values in a wave — the top
// byte of float bit patterns, and the byte below it (NDC z crowds towards 1: on the 50 M scene 64 lanes hit two or three
// counters, and LDS atomics on one address are served one lane at a time: 150 M of them took 0.16 ms).
__device__ __forceinline__ void wave_count_digit(uint32_t* counters, uint32_t d, bool vis) {
    unsigned long long todo = __ballot(vis);
    while (todo) {
        const uint32_t val = (uint32_t)__builtin_amdgcn_readlane((int)d, __ffsll((long long)todo) - 1);
        const unsigned long long same = __ballot(vis && d == val) & todo;
        if ((threadIdx.x & (kWave - 1)) == 0) atomicAdd(&counters[val], (uint32_t)__popcll(same));
        todo &= ~same;
    }
}

// ---- visible keys first: stable compaction of the keys that are not the 0xFFFFFFFF sentinel --------
constexpr int kCompactThreads = 256, kCompactRows = 16, kCompactChunk = kCompactThreads * kCompactRows;
constexpr int kByte2Copies = 8;
static_assert(kCompactChunk == 4096, "the scan of tilesTouched (scan.hip) counts the visible Gaussians per 4096 elements for this compaction");

__global__ __launch_bounds__(kCompactThreads) void visible_count_kernel(const uint32_t* __restrict__ keys, uint32_t n,
                                                                        uint32_t* __restrict__ partial) {
    const uint32_t base = blockIdx.x * kCompactChunk;
    uint32_t c = 0;
#pragma unroll
    for (int r = 0; r < kCompactRows; ++r) {
        const uint32_t e = base + (uint32_t)r * kCompactThreads + threadIdx.x;
        c += (e < n && keys[e] != 0xFFFFFFFFu) ? 1u : 0u;
    }
    __shared__ uint32_t s_w[kCompactThreads / kWave];
    uint32_t v = c;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += (uint32_t)__shfl_xor((int)v, off, kWave);
    if ((threadIdx.x & (kWave - 1)) == 0) s_w[threadIdx.x / kWave] = v;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

// exclusive prefix of the per-chunk counts, in place; the total (visible keys) goes to *total_out
__global__ __launch_bounds__(1024) void visible_scan_kernel(uint32_t* __restrict__ partial, uint32_t chunks,
                                                            uint32_t* __restrict__ total_out) {
    __shared__ uint32_t s_ws[16];
    const uint32_t per = (chunks + 1023) / 1024;
    const uint32_t a = min(chunks, threadIdx.x * per), z = min(chunks, a + per);
    uint32_t s = 0;
    for (uint32_t i = a; i < z; ++i) s += partial[i];
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    uint32_t incl = s;
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
        const uint32_t o = __shfl_up(incl, off, kWave);
        if (lane >= off) incl += o;
    }
    if (lane == kWave - 1) s_ws[wave] = incl;
    __syncthreads();
    uint32_t running = incl - s;
    for (int w = 0; w < wave; ++w) running += s_ws[w];
    for (uint32_t i = a; i < z; ++i) {
        const uint32_t c = partial[i];
        partial[i] = running;
        running += c;
    }
    if (threadIdx.x == 1023) *total_out = running;
}

// writes the visible (key, index) pairs in index order and counts the four 8-bit digits of every
// visible key (the histograms of the four sort passes) on the way
// rect_by_index / out_r (may be null): the visible Gaussians' packed rectangles are compacted with the pairs — here, in
// index order, that read is coalesced; they then travel through the depth passes as the keys' second value.
__global__ __launch_bounds__(kCompactThreads) void visible_compact_kernel(const uint32_t* __restrict__ keys, uint32_t n,
                                                                          const uint32_t* partial,
                                                                          const uint32_t* __restrict__ rect_by_index,
                                                                          uint32_t* __restrict__ out_k, uint32_t* __restrict__ out_v,
                                                                          uint32_t* __restrict__ out_r,
                                                                          uint32_t* __restrict__ hist, const DepthSide side) {
    constexpr int kCompactWaves = kCompactThreads / kWave;
    static_assert(kCompactRows * kCompactWaves == kWave, "one wave scans the (row, wave) counts");
    // (digit counters: bytes 0, 1, 3, then kByte2Copies copies of byte 2's — NDC z crowds towards 1 and the lanes of a wave
    // share two or three values of that byte; LDS atomics on one address are served a lane at a time: with one copy the
    // 50 M scene's 46 M keys spent 75 us there, with a copy per eighth of the wave a tenth of it)
    __shared__ uint32_t lds[(3 + kByte2Copies) * 256];
    __shared__ uint32_t s_off[kCompactRows * kCompactWaves];       // (row, wave): keys of that row in that wave, then where they go
    for (int i = threadIdx.x; i < (3 + kByte2Copies) * 256; i += kCompactThreads) lds[i] = 0;
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    // A workgroup takes several chunks (chunk = blockIdx.x, += gridDim.x) and adds its digit counts to the global
    // histograms ONCE: with a workgroup per chunk the 1 425 workgroups of the bench frame each sent 1 024 atomics to the same
    // 1 024 words, and same-address atomics are served one at a time (about 17 ns each on this part).
    const uint32_t chunks = (n + kCompactChunk - 1) / kCompactChunk;
    // The side way (DepthSide): only the keys with the main top byte are compacted, with the offsets counted for them
    const bool side_on = side.words != nullptr && side.words[0] != 0u;
    if (side_on) partial = side.main_partial;
    for (uint32_t chunk = blockIdx.x; chunk < chunks; chunk += gridDim.x) {
    const uint32_t base = chunk * kCompactChunk;
    __syncthreads();                                     // (s_off of the previous chunk has been read by everybody)
    // All rows are loaded before anything is counted (one round trip), and the output offsets of all (row, wave)
    // pieces come from ONE 64-lane scan: two barriers per workgroup (a barrier pair per row before: 32).
    uint32_t k[kCompactRows], rect[kCompactRows];
#pragma unroll
    for (int r = 0; r < kCompactRows; ++r) {
        const uint32_t e = base + (uint32_t)r * kCompactThreads + threadIdx.x;
        k[r] = (e < n) ? keys[e] : 0xFFFFFFFFu;
    }
    if (side_on) {
        // the few other visible keys go to the side list (any order: depth_side_kernel ranks them) and leave this stream
#pragma unroll
        for (int r = 0; r < kCompactRows; ++r) {
            if (k[r] != 0xFFFFFFFFu && (k[r] >> 24) != side.main_top) {
                const uint32_t e = base + (uint32_t)r * kCompactThreads + threadIdx.x;
                const uint32_t slot = atomicAdd(&side.words[1], 1u);
                if (slot < side.capacity) {              // (always: the scan counted them before it chose this way)
                    side.keys[slot] = k[r];
                    side.vals[slot] = e;
                    if (side.rects) side.rects[slot] = rect_by_index ? rect_by_index[e] : 0u;
                }
                if ((k[r] >> 24) < side.main_top) atomicAdd(&side.words[2], 1u);
                k[r] = 0xFFFFFFFFu;
            }
        }
    }
    if (rect_by_index) {
        // (unconditional: the preprocess writes a rectangle word for every Gaussian, 0 for the culled ones)
#pragma unroll
        for (int r = 0; r < kCompactRows; ++r) {
            const uint32_t e = base + (uint32_t)r * kCompactThreads + threadIdx.x;
            rect[r] = (e < n) ? rect_by_index[e] : 0u;
        }
    }
    unsigned long long m[kCompactRows];
    uint32_t mine = 0;                                   // lane r: this wave's visible keys in row r
#pragma unroll
    for (int r = 0; r < kCompactRows; ++r) {
        m[r] = __ballot(k[r] != 0xFFFFFFFFu);
        mine = (lane == r) ? (uint32_t)__popcll(m[r]) : mine;
    }
    if (lane < kCompactRows) s_off[lane * kCompactWaves + wave] = mine;
    __syncthreads();
    if (wave == 0) {
        const uint32_t c = s_off[lane];
        uint32_t incl = c;
#pragma unroll
        for (int off = 1; off < kWave; off <<= 1) {
            const uint32_t o = __shfl_up(incl, off, kWave);
            if (lane >= off) incl += o;
        }
        s_off[lane] = partial[chunk] + incl - c;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kCompactRows; ++r) {
        const uint32_t e = base + (uint32_t)r * kCompactThreads + threadIdx.x;
        const bool vis = k[r] != 0xFFFFFFFFu;
        if (vis) {
            const uint32_t pos = s_off[r * kCompactWaves + wave] +
                                 __builtin_amdgcn_mbcnt_hi((uint32_t)(m[r] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m[r], 0u));
            out_k[pos] = k[r];
            out_v[pos] = e;
            if (out_r) out_r[pos] = rect[r];
            atomicAdd(&lds[k[r] & 255u], 1u);
            atomicAdd(&lds[256 + ((k[r] >> 8) & 255u)], 1u);
            atomicAdd(&lds[768 + (lane & (kByte2Copies - 1)) * 256 + ((k[r] >> 16) & 255u)], 1u);
        }
        // (the top byte of float bit patterns takes few distinct values in a wave)
        wave_count_digit(lds + 512, k[r] >> 24, vis);
    }
    }
    __syncthreads();
    // hist: [byte 0 | byte 1 | byte 2 | byte 3] x 256
    for (int i = threadIdx.x; i < 4 * 256; i += kCompactThreads) {
        uint32_t c;
        if (i < 512) c = lds[i];
        else if (i < 768) { c = 0; for (int r = 0; r < kByte2Copies; ++r) c += lds[768 + r * 256 + (i - 512)]; }
        else c = lds[512 + (i - 768)];
        if (c) atomicAdd(&hist[i], c);
    }
}
// ---- the side way: the few visible keys whose top byte is not the main one -----------------------------------------
// Depth keys are float bits of NDC z. Nearly every visible Gaussian has z in [0.5, 1): top byte 0x3F; the fourth sort
// pass exists for the others — 23 of 3 043 608 on the bench frame, one of 5.6 M from outside the cloud — and moved every key
// for them (35 - 52 us). When they are few (the scan counts them: scan.hip) the compaction leaves them out of the stream,
// three passes sort the rest, and this kernel ranks the side list by (key, index) — the order a stable sort gives — and
// puts the keys below the main top byte in front of the sorted stream (the arrays have room there) and those above behind.
__global__ __launch_bounds__(1024) void depth_side_kernel(const DepthSide side, uint32_t m, uint32_t m_lo, uint32_t main_count,
                                                          uint32_t* __restrict__ out_k, uint32_t* __restrict__ out_v,
                                                          uint32_t* __restrict__ out_r) {
    __shared__ uint32_t s_k[kDepthSideMax], s_v[kDepthSideMax];
    const uint32_t t = threadIdx.x;
    uint32_t key = 0, val = 0, rect = 0;
    if (t < m) { key = side.keys[t]; val = side.vals[t]; rect = side.rects ? side.rects[t] : 0u; s_k[t] = key; s_v[t] = val; }
    __syncthreads();
    if (t >= m) return;
    const bool low = (key >> 24) < side.main_top;
    uint32_t rank = 0;                           // keys of the same side in front of this one
    for (uint32_t j = 0; j < m; ++j) {
        const uint32_t kj = s_k[j], vj = s_v[j];
        const bool same = ((kj >> 24) < side.main_top) == low;
        rank += (same && (kj < key || (kj == key && vj < val))) ? 1u : 0u;
    }
    // out_*: the sorted stream's first element; low keys at [-m_lo, 0), high keys at [main_count, main_count + m - m_lo)
    const long long at = low ? (long long)rank - (long long)m_lo : (long long)main_count + (long long)rank;
    out_k[at] = key;
    out_v[at] = val;
    if (out_r) out_r[at] = rect;
}
// The digit counts visible_compact_kernel takes on its way, without the compaction: for scenes beyond 16 M Gaussians, whose
// first depth pass reads the per-Gaussian keys itself (onesweep_kernel, DROP). A key takes part if it is not the sentinel and,
// when the side way is taken, has the main top byte. A workgroup walks many chunks and adds to the global counters once.
__global__ __launch_bounds__(256) void depth_hist_kernel(const uint32_t* __restrict__ keys, uint32_t n, uint32_t* __restrict__ hist,
                                                         const DepthSide side) {
    __shared__ uint32_t lds[(3 + kByte2Copies) * 256];   // bytes 0, 1, 3, then the copies of byte 2's counters (see visible_compact_kernel)
    __shared__ uint32_t s_below;
    for (int i = threadIdx.x; i < (3 + kByte2Copies) * 256; i += 256) lds[i] = 0;
    if (threadIdx.x == 0) s_below = 0;
    __syncthreads();
    const bool side_on = side.words != nullptr && side.words[0] != 0u;
    const int lane = threadIdx.x & (kWave - 1);
    const uint32_t vecs = (n + 3u) / 4u;
    uint32_t top_main = 0;                               // (wave-uniform) keys of this wave with the main top byte, side way
    constexpr int kInFlight = 4;                         // 16-byte loads per lane and round (a lane with one load in flight waits on latency: 170 us for 200 MB)
    const uint32_t stride = gridDim.x * 256u;
    for (uint32_t v0 = blockIdx.x * 256u + threadIdx.x; v0 < ((vecs + 255u) & ~255u); v0 += kInFlight * stride) {      // (whole workgroups take a round or leave: the ballots below)
        uint4 q[kInFlight];
#pragma unroll
        for (int u = 0; u < kInFlight; ++u) {
            const uint32_t v = v0 + (uint32_t)u * stride;
            q[u] = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
            if (4ull * v + 3ull < n) {
                q[u] = *reinterpret_cast<const uint4*>(keys + 4u * (size_t)v);
            } else if (4ull * v < n) {
                uint32_t t[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
                for (int j = 0; j < 4; ++j) if (4ull * v + (uint32_t)j < n) t[j] = keys[4u * (size_t)v + j];
                q[u] = make_uint4(t[0], t[1], t[2], t[3]);
            }
        }
#pragma unroll
        for (int u = 0; u < kInFlight; ++u) {
            const uint32_t k[4] = {q[u].x, q[u].y, q[u].z, q[u].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool vis = k[j] != 0xFFFFFFFFu && (!side_on || (k[j] >> 24) == side.main_top);
                // (side way: the keys that will go to the side list and sort in FRONT of the stream — side.words[2], which the host reads)
                if (side_on && k[j] != 0xFFFFFFFFu && (k[j] >> 24) < side.main_top) atomicAdd(&s_below, 1u);
                if (vis) {
                    atomicAdd(&lds[k[j] & 255u], 1u);
                    atomicAdd(&lds[256 + ((k[j] >> 8) & 255u)], 1u);
                    atomicAdd(&lds[768 + (lane & (kByte2Copies - 1)) * 256 + ((k[j] >> 16) & 255u)], 1u);
                }
                // (on the side way every key that takes part has the main top byte)
                if (side_on) top_main += (uint32_t)__popcll(__ballot(vis));
                else wave_count_digit(lds + 512, k[j] >> 24, vis);
            }
        }
    }
    if (lane == 0 && top_main) atomicAdd(&lds[512 + side.main_top], top_main);
    __syncthreads();
    for (int i = threadIdx.x; i < 4 * 256; i += 256) {
        uint32_t c;
        if (i < 512) c = lds[i];
        else if (i < 768) { c = 0; for (int r = 0; r < kByte2Copies; ++r) c += lds[768 + r * 256 + (i - 512)]; }
        else c = lds[512 + (i - 768)];
        if (c) atomicAdd(&hist[i], c);
    }
    if (threadIdx.x == 0 && s_below) atomicAdd(&side.words[2], s_below);
}
}  // namespace

int launch_depth_side(const DepthSide& side, uint32_t m, uint32_t m_lo, uint32_t main_count, uint32_t* out_k, uint32_t* out_v,
                      uint32_t* out_r, hipStream_t stream) {
    if (m == 0) return GSR_OK;
    if (m > kDepthSideMax || m_lo > m) return GSR_ERR_INTERNAL;
    hipLaunchKernelGGL(depth_side_kernel, dim3(1), dim3(1024), 0, stream, side, m, m_lo, main_count, out_k, out_v, out_r);
    GSR_LAUNCH_CHECK("depth_side_kernel");
    return GSR_OK;
}

size_t depth_compact_scratch_bytes(size_t n) { return align_up(((n + kCompactChunk - 1) / kCompactChunk) * sizeof(uint32_t), 128); }

// Depth keys of the visible Gaussians first (stable: index order), their count, the digit histograms of
// the four sort passes, and (top_digits) the number of distinct top-byte digits: with at most one the
// fourth pass would move nothing. info[0] = top_digits, info[1] = visible count (device words); info[4] is zeroed
// (the frame's non-empty-tile counter; info must hold at least five words).
int sort_u32_prepare(const uint32_t* keys_in, uint32_t n, uint32_t* out_k, uint32_t* out_v, uint32_t* partial,
                     const SweepScratch* sc4, uint32_t* info, hipStream_t stream, bool offsets_ready,
                     const uint32_t* rect_by_index, uint32_t* out_r, uint32_t* host_top, const DepthSide* side) {
    if (n == 0) return GSR_OK;
    const uint32_t chunks = (n + kCompactChunk - 1) / kCompactChunk;
    if (!offsets_ready) {
        hipLaunchKernelGGL(visible_count_kernel, dim3(chunks), dim3(kCompactThreads), 0, stream, keys_in, n, partial);
        GSR_LAUNCH_CHECK("visible_count_kernel");
        hipLaunchKernelGGL(visible_scan_kernel, dim3(1), dim3(1024), 0, stream, partial, chunks, info + 1);
        GSR_LAUNCH_CHECK("visible_scan_kernel");
    }
    if ((rect_by_index == nullptr) != (out_r == nullptr)) return GSR_ERR_INVALID_ARG;
    // (chunks per workgroup: bench frame, 1 425 chunks: 41 -> 32 us with two, no better with three to six; 50 M Gaussians,
    // 12 208 chunks: 288 -> 205 us with two to six)
    const uint32_t per_wg = chunks >= 4096u ? 4u : 2u;
    DepthSide no_side;
    hipLaunchKernelGGL(visible_compact_kernel, dim3((chunks + per_wg - 1) / per_wg), dim3(kCompactThreads), 0, stream, keys_in, n, partial, rect_by_index,
                       out_k, out_v, out_r, sc4[0].hist, side ? *side : no_side);
    GSR_LAUNCH_CHECK("visible_compact_kernel");
    hipLaunchKernelGGL(top_digit_count_kernel, dim3(1), dim3(256), 0, stream, sc4[0].hist + 3 * 256, info, host_top,
                       side ? side->words : nullptr);
    GSR_LAUNCH_CHECK("top_digit_count_kernel");
    return GSR_OK;
}

// The same without the compaction (scenes beyond 16 M Gaussians): only the digit counts of the keys that take part; the first
// pass then reads keys_in itself (sort_u32_passes with `drop_side`).
int sort_u32_prepare_counts(const uint32_t* keys_in, uint32_t n, const SweepScratch* sc4, uint32_t* info, hipStream_t stream,
                            uint32_t* host_top, const DepthSide* side) {
    if (n == 0) return GSR_OK;
    DepthSide no_side;
    const uint32_t wgs = std::min<uint32_t>((n + 4095u) / 4096u, 2048u);
    hipLaunchKernelGGL(depth_hist_kernel, dim3(wgs), dim3(256), 0, stream, keys_in, n, sc4[0].hist, side ? *side : no_side);
    GSR_LAUNCH_CHECK("depth_hist_kernel");
    hipLaunchKernelGGL(top_digit_count_kernel, dim3(1), dim3(256), 0, stream, sc4[0].hist + 3 * 256, info, host_top,
                       side ? side->words : nullptr);
    GSR_LAUNCH_CHECK("top_digit_count_kernel");
    return GSR_OK;
}

// Passes [first, last) of the stable sort of n (key, value) u32 pairs: in -> a -> b -> a -> b. After P
// passes the result is in (a_k, a_v) if P is odd, else in (b_k, b_v).
// n_dev (may be null): the true key count on the device, n then being an upper bound that only sizes the grids.
// second_in / a_s / b_s (all null, or none): a second value per key that takes the same path (in -> a -> b -> ...).
// rec_a / rec_b (both null, or none; with second values and first == 0 only): room for 3 n words each — the triples then travel
// BETWEEN the passes as 12-byte records (in -> rec_a -> rec_b -> rec_a ...), and the LAST pass of the call writes the three
// arrays of its turn (a_* / b_*) as without them. rec_a may overlap the arrays of the last pass's turn (dead by then), rec_b not.
int sort_u32_passes(const uint32_t* keys_in, const uint32_t* vals_in, uint32_t n, uint32_t* a_k, uint32_t* a_v, uint32_t* b_k,
                    uint32_t* b_v, const SweepScratch* sc4, int first, int last, hipStream_t stream, const uint32_t* n_dev,
                    const uint32_t* second_in, uint32_t* a_s, uint32_t* b_s, const DepthSide* drop_side, uint32_t* rec_a,
                    uint32_t* rec_b) {
    if (n == 0) return GSR_OK;
    if ((rec_a == nullptr) != (rec_b == nullptr) || (rec_a && (!second_in || first != 0))) return GSR_ERR_INVALID_ARG;
    // drop_side (pass 0 only): keys_in is the per-Gaussian array of n keys with sentinels (sort_u32_prepare_counts); vals_in must
    // be null, second_in the per-Gaussian second values, n_dev the number of keys that take part
    if (drop_side && (vals_in || !second_in || !n_dev)) return GSR_ERR_INVALID_ARG;
    for (int p = first; p < last; ++p) {
        const uint32_t* src_k = (p == 0) ? keys_in : ((p % 2 == 1) ? a_k : b_k);
        const uint32_t* src_v = (p == 0) ? vals_in : ((p % 2 == 1) ? a_v : b_v);
        uint32_t* dst_k = (p % 2 == 0) ? a_k : b_k;
        uint32_t* dst_v = (p % 2 == 0) ? a_v : b_v;
        const uint32_t* src_s = second_in ? ((p == 0) ? second_in : ((p % 2 == 1) ? a_s : b_s)) : nullptr;
        uint32_t* dst_s = second_in ? ((p % 2 == 0) ? a_s : b_s) : nullptr;
        int rec = 0;
        if (rec_a) {
            if (p > 0) { rec |= kRecIn; src_k = (p % 2 == 1) ? rec_a : rec_b; src_v = nullptr; }
            if (p < last - 1) { rec |= kRecOut; dst_k = (p % 2 == 0) ? rec_a : rec_b; }
        }
        DigitSpec spec;
        spec.mode = kDigitBits; spec.shift = 8 * p; spec.nbins = 256; spec.grid_x = 1; spec.inv_grid_x = 1.0f;
        SweepScratch sc = sc4[p];
        sc.error_word = sc4[0].error_word;
        sc.error_value = sc4[0].error_value;
        int rc;
        if (p == 0 && drop_side) {
            DropSpec drop;
            drop.n_out = n_dev;
            drop.side = *drop_side;
            rc = launch_pass<uint32_t>(src_k, nullptr, dst_k, dst_v, n, spec, sc4[0].hist, sc, stream, true, nullptr, src_s, dst_s, &drop, rec);
        } else {
            rc = launch_pass<uint32_t>(src_k, src_v, dst_k, dst_v, n, spec, sc4[0].hist + 256 * p, sc, stream, true, n_dev, src_s, dst_s,
                                       nullptr, rec);
        }
        if (rc != GSR_OK) return rc;
    }
    return GSR_OK;
}

// ---- generic entry: stable sort of (u64, u32) pairs on key bits [0, end_bit) -------------
size_t sort_temp_bytes(size_t n) {
    return align_up(n * sizeof(uint64_t), 128) + align_up(n * sizeof(uint32_t), 128) + sweep_scratch_bytes(n);
}

int launch_sort_pairs(const uint64_t* keys_in, uint64_t* keys_out, const uint32_t* values_in, uint32_t* values_out,
                      size_t n, int begin_bit, int end_bit, char* temp, hipStream_t stream, uint32_t* error_word, uint32_t error_value) {
    if (n == 0) return GSR_OK;
    if (n >= 0xFFFFFFFFull) return GSR_ERR_TOO_LARGE;
    if (end_bit > 64) end_bit = 64;
    if (begin_bit < 0) begin_bit = 0;
    if (end_bit <= begin_bit) end_bit = begin_bit + 1;
    const int passes = (end_bit - begin_bit + 7) / 8;
    uint64_t* tmp_k = reinterpret_cast<uint64_t*>(temp);
    uint32_t* tmp_v = reinterpret_cast<uint32_t*>(temp + align_up(n * sizeof(uint64_t), 128));
    SweepScratch sc = carve_sweep_scratch(temp + align_up(n * sizeof(uint64_t), 128) + align_up(n * sizeof(uint32_t), 128), n);
    if (error_word) { sc.error_word = error_word; sc.error_value = error_value; }          // (the caller's word: gsr_forward hands in this call's slot of pinned host memory, zeroed at the top of the call)
    else GSR_HIP_TRY(hipMemsetAsync(sc.error_word, 0, sizeof(uint32_t), stream));
    int rc = histogram_bits_u64(keys_in, n, begin_bit, end_bit, sc.hist, stream);
    if (rc != GSR_OK) return rc;
    const uint64_t* src_k = keys_in;
    const uint32_t* src_v = values_in;
    for (int p = 0; p < passes; ++p) {
        const bool to_out = ((passes - 1 - p) % 2) == 0;     // the last pass lands in the outputs
        uint64_t* dst_k = to_out ? keys_out : tmp_k;
        uint32_t* dst_v = to_out ? values_out : tmp_v;
        DigitSpec spec;
        spec.mode = kDigitBits;
        spec.shift = begin_bit + 8 * p;
        spec.nbins = 1u << std::min(8, end_bit - spec.shift);
        spec.grid_x = 1;
        spec.inv_grid_x = 1.0f;
        rc = sweep_pass_u64(src_k, src_v, dst_k, dst_v, (uint32_t)n, spec, sc.hist + 256 * p, sc, stream);
        if (rc != GSR_OK) return rc;
        src_k = dst_k;
        src_v = dst_v;
    }
    return GSR_OK;
}

}  // namespace gsr
