// Internal interface of the two-pass depth order (depth_buckets.hip).
#pragma once
#include <algorithm>

#include "gsr_common.hpp"

namespace gsr {

constexpr uint32_t kDepthSampleStride = 256;    // one sample per this many Gaussians (16 per 4096-element tile of the scan)
constexpr uint32_t kDepthOversample = 8;        // samples per bucket
constexpr uint32_t kDepthBucketCap = 8192;      // entries a bucket's region (and the LDS of its sort) holds: 4 x the mean
constexpr uint32_t kDepthMaxBuckets = 4096;

// Which sample the scan takes out of group g (256 consecutive Gaussians): a hash of the group number, so that no
// regularity of the input order meets a regular stride.
__host__ __device__ inline uint32_t depth_sample_index(uint32_t g) {
    return g * kDepthSampleStride + ((g * 2654435761u) >> 24);
}

struct DepthBuckets {
    uint32_t* words;          // [0] buckets, [1] failure flag, [2] valid samples         } zeroed before every frame
    uint32_t* counts;         // [kDepthMaxBuckets] entries per bucket                    } (depth_buckets_cleared_bytes from words)
    unsigned long long* samples;     // [groups] depth bits << 32 | index, or ~0
    unsigned long long* splitters;   // [buckets - 1]
    char* regions;            // [max_buckets][kDepthBucketCap] x 16 bytes: {depth bits, index, packed rectangle, 0}
    uint32_t groups, max_buckets;
    size_t bytes;
};
bool depth_buckets_supported(size_t n);
size_t depth_buckets_cleared_bytes();
size_t depth_buckets_scratch_bytes(size_t n);
DepthBuckets carve_depth_buckets(char* base, size_t n);
int launch_depth_bucket_scatter(const DepthBuckets& d, const uint32_t* depth_key, const uint32_t* rect_by_index, uint32_t n,
                                const uint32_t* visible_dev, uint32_t* host_fail, hipStream_t stream);
int launch_depth_bucket_sort(const DepthBuckets& d, uint32_t n, uint32_t* out_k, uint32_t* out_v, uint32_t* out_r,
                             hipStream_t stream);

}  // namespace gsr
