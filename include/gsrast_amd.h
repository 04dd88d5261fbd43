/* gsrast_amd.h — C ABI of the MI355X (gfx950) forward Gaussian-splat rasterizer.
 *
 * Drop-in boundary for the reference's rasterizer entry point
 *   gscuda::forward            apps/gsrast/gscuda/GSCuda.cuh:103-126 (def. GSCuda.cu:695-811)
 *   ≅ CudaRasterizer::Rasterizer::forward (apps/gsrast/GSGaussians.cpp:18-23)
 * and for the chunk-layout helpers its callers re-derive pointers with
 *   gscuda::required<T>, gs::{Geometry,Image,Binning}State::fromChunk
 *                              apps/gsrast/gscuda/AuxBuffer.cuh:8-14,38-76, AuxBuffer.cu:13-21,44-89
 *
 * The reference interface is a C++ ABI (by-value std::function, namespaces); this header is
 * the plain-C core a binding targets. include/gscuda_shim.hpp re-creates the exact C++
 * signature on top of it. All pointers named "device" are HIP device pointers owned by
 * the caller. Everything a frame needs lives in the caller's chunks; what the library itself allocates, once per host
 * thread and device, is listed here in full: 1 KB of pinned host memory (the numRendered read-back and the calls' error
 * words), a second stream of the lowest priority and a handful of events, an 8-byte device counter (GSR_FLAG_COUNT_STAGED
 * only), and per tile history (below: one per stream the thread calls on, at most eight, or the caller's own
 * gsr_tile_history) 384 KB of device memory and 64 bytes of pinned host memory. None of it influences any output. The
 * reference owns nothing (GSCuda.cu:723-784: caller chunks only), so all of it can be given back: gsr_thread_release()
 * frees what the CALLING thread's calls have allocated, and the library does the same by itself when a thread that called
 * it ends — a host that renders from short-lived worker threads keeps nothing behind.
 */
#ifndef GSRAST_AMD_H
#define GSRAST_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GSR_VERSION 1

/* Error codes (gsr_forward's return value; the reference returns void and its caller
 * polls the runtime's sticky error, apps/gsrast/CudaBuffer.hpp:8-12). */
enum {
    GSR_OK = 0,
    GSR_ERR_INVALID_ARG = 1,   /* null pointer / non-positive size / bad struct_size      */
    GSR_ERR_ALLOC = 2,         /* an allocator callback returned NULL                      */
    GSR_ERR_HIP = 3,           /* a HIP runtime call or kernel launch failed               */
    GSR_ERR_NO_DEVICE = 4,     /* no gfx950 device visible                                 */
    GSR_ERR_TOO_LARGE = 5,     /* sum of tiles touched >= 2^32 - 1: does not fit the reference's u32 offsets (checked
                                  on the un-wrapped 64-bit total before the binning chunk is requested) */
    GSR_ERR_INTERNAL = 6,      /* a bounded device-side wait expired (see gsr_poll_async_error) */
    GSR_ERR_STALE_RECEIPT = 7  /* gsr_poll_async_error: the receipt's error slot belongs to a later call (64 calls on): unknown */
};

/* Chunk allocator: replaces std::function<char*(size_t)> (GSCuda.cuh:103-105). Must
 * return device memory of at least `bytes` bytes, 128-byte aligned or better. */
typedef char* (*gsr_alloc_fn)(void* user, size_t bytes);

/* gs::GeometryState (AuxBuffer.cuh:38-54); field order = carve order (AuxBuffer.cu:44-63). */
typedef struct gsr_geometry_state {
    uint32_t* tiles_touched;   /* u32[N]                                   */
    size_t    scan_size;       /* bytes of scanning_space                  */
    uint32_t  num_rendered;    /* host copy of point_offsets[N-1]          */
    char*     scanning_space;  /* scan scratch (this library's own layout) */
    float*    depths;          /* f32[N]  NDC z                            */
    uint8_t*  clamped;         /* bool[3N], never written (as reference)   */
    int32_t*  internal_radii;  /* i32[N]                                   */
    float*    means2D;         /* vec2[N]                                  */
    float*    cov3D;           /* f32[6N]                                  */
    float*    conic_opacity;   /* vec4[N]                                  */
    float*    rgb;             /* vec3[N], 12-byte stride                  */
    uint32_t* point_offsets;   /* u32[N] inclusive scan of tiles_touched   */
} gsr_geometry_state;

/* gs::ImageState (AuxBuffer.cuh:56-63). `size` = width*height. */
typedef struct gsr_image_state {
    uint32_t* ranges;          /* uvec2[size]; only the first ceil(W/16)*ceil(H/16) are used */
    uint32_t* n_contrib;       /* u32[size]                                */
    float*    accum_alpha;     /* f32[size] final transmittance            */
} gsr_image_state;

/* gs::BinningState (AuxBuffer.cuh:65-75). `size` = numRendered. */
typedef struct gsr_binning_state {
    uint64_t* keys_unsorted;   /* u64[R] (tile << 32) | depth bits         */
    uint64_t* keys;            /* u64[R] sorted                            */
    uint32_t* values_unsorted; /* u32[R] Gaussian index                    */
    uint32_t* values;          /* u32[R] sorted                            */
    size_t    sorting_size;    /* bytes of sorting_space                   */
    char*     sorting_space;   /* radix-sort scratch                       */
} gsr_binning_state;

/* fromChunk: carve the sub-arrays out of `chunk`; returns the advanced chunk pointer
 * (the reference advances a char*& in place). */
char* gsr_geometry_from_chunk(char* chunk, int num_gaussians, gsr_geometry_state* out);
char* gsr_image_from_chunk(char* chunk, int size, gsr_image_state* out);
char* gsr_binning_from_chunk(char* chunk, size_t size, gsr_binning_state* out);
/* required<T>(n): fromChunk on a null base (AuxBuffer.cuh:8-14). */
size_t gsr_required_geometry(int num_gaussians);
size_t gsr_required_image(int size);
size_t gsr_required_binning(size_t size);

/* Per-stage device times of the last profiled gsr_forward call, in milliseconds. */
enum {
    GSR_STAGE_PREPROCESS = 0,   /* preprocess kernel                                             */
    GSR_STAGE_SCAN = 1,         /* inclusive scan of tiles touched (3 small kernels)             */
    GSR_STAGE_DEPTH_ORDER = 2,  /* per-Gaussian depth sort + N-sized binning preparation         */
    GSR_STAGE_DUPLICATE = 3,    /* key emission kernel                                           */
    GSR_STAGE_SORT_PASS1 = 4,   /* block plan: unit counts + prefixes; else unused               */
    GSR_STAGE_SORT_PASS2 = 5,   /* onesweep kernel, tile-row digit (remaining passes)            */
    GSR_STAGE_RANGES = 6,       /* clear + tile ranges kernel                                    */
    GSR_STAGE_BLEND = 7,        /* blend kernel                                                  */
    GSR_NUM_STAGES = 8
};

#define GSR_FLAG_PROFILE 0x1u   /* record HIP events around every stage into stage_ms       */
#define GSR_FLAG_COUNT_STAGED 0x2u /* count records staged by the blend stage (R_f) into records_staged */
/* Semantics profile of the UPSTREAM rasterizer (graphdeco-inria/diff-gaussian-rasterization) instead
 * of the reference's gscuda variant (SURVEY.md §8a divergence table D1-D12, §8f-2): SH up to degree
 * `sh_dims` on shs laid out [N][16][3], view-space depth keys, cull at view z <= 0.2, half-pixel
 * centre, radius-based square rectangles (`rects` ignored), separate focal lengths, transmittance
 * cut-off 1e-4, R == 1 renders, R == 0 still writes the background. cam_pos is read. Parity of this
 * profile is unpinned (no upstream source in the reference tree). */
#define GSR_FLAG_SEMANTICS_INRIA 0x4u
/* Binning plan (both produce bit-identical sorted keys / values / ranges; default: chosen per frame: the block plan from 6
 * instances (numRendered) per visible Gaussian up, or when an eighth of the instances belongs to splats of 256 tiles and more —
 * the few hundred background splats of a trained scene seen from outside — whatever the average). PLAN_SORT: column-major key
 * emission + one onesweep radix pass;
 * keysUnsorted / valuesUnsorted then hold the reference's pairs in (tile column, depth) order.
 * PLAN_BLOCKS: the sorted lists are written directly by tile-block owners, no R-sized sort;
 * keysUnsorted / valuesUnsorted then hold that plan's block lists (scratch, as sortingSpace is). */
#define GSR_FLAG_PLAN_SORT 0x8u
#define GSR_FLAG_PLAN_BLOCKS 0x10u
/* Block plan: the blend can run on a second stream beside the emission of the sorted lists — one is bound by vector issue, the
 * other by the HBM write path (or, on light frames, by nothing: a wave per unit) — reading the block lists instead of the sorted
 * ones. Same results; the call's work is complete, as always, when `stream` is. It pays where the blend is the shorter of the
 * two (bench frame 1.34 -> 1.22 ms, 4K 4.21 -> 3.85 ms; faint splats, blend three times the emission: 3.10 -> 3.16), and each
 * kernel runs longer while they share the chip. By DEFAULT the library decides per call, from 16 instances per visible Gaussian
 * up, from the call's tile history (see GSR_FLAG_NO_TILE_HISTORY; without one, or while its frames do not resemble each other:
 * serial) and this call's R: overlapped when the blend — the tiles' times spread over the chip, never less than the longest
 * tile, and 1 + 10 V / R times that if they were measured on a blend fed from the sorted lists — is expected to end before an
 * emission of 12 R bytes at 4 TB/s does (twice that on frames that are block-fed either way; on light frames: before 0.14 ms);
 * plan_used then carries GSR_PLAN_EMIT_OVERLAPPED. GSR_FLAG_OVERLAP_EMIT forces it on (block plan), GSR_FLAG_SERIAL_EMIT off: one
 * kernel after the other on the caller's stream, per-kernel times those of the kernels alone; a serial blend reads the sorted
 * lists below 48 instances per visible Gaussian and the block lists from there on. (`profiles/r05_trained_like.txt`: the
 * library's choice against every forced one on 240 poses of two scene families.) */
#define GSR_FLAG_OVERLAP_EMIT 0x20u
#define GSR_FLAG_SERIAL_EMIT 0x100u
/* (Also by default, gscuda semantics without colors_precomp: geomState.rgb is written by a kernel of its own on the
 * library's second stream — a strided read nothing needs before the blend, 0.10 ms of the bench frame's preprocess. Up to
 * 16 M Gaussians right behind the preprocess, while the scan and the depth sort run (0.02-0.03 ms more there), and the
 * call's stream waits for it before the blend; beyond,
 * where the depth sort is bound by HBM itself, beside the blend, which takes a record's colour straight from `shs` meanwhile
 * (50 M Gaussians: 6.3 -> 5.9 ms). The caller's stream has waited for it on every way out of the call. Same bits.
 * GSR_FLAG_SERIAL_EMIT keeps the colours in the preprocess kernel as well.) */
/* Forward-only callers: under the block plan the blend can be fed from the block lists without reading the sorted
 * keys / values (it is by default on frames of 48 or more instances per visible Gaussian), and no caller of the reference reads BinningState (GSGaussians.cpp:214-219 maps GeometryState
 * only). With this flag such a call skips writing them (12 R bytes): BinningState.keys / values are then left
 * UNWRITTEN except for values[0] = GSR_LISTS_SKIPPED_STAMP, plan_used carries GSR_PLAN_LISTS_SKIPPED, and a gsr_backward
 * call that is given this call's receipt (any thread; chunks untouched) walks the tile lists out of the block lists.
 * Pixels, ranges, finalT, nContrib and numRendered are unchanged. Ignored under the other plans, whose blend
 * reads the sorted list. Default off: the reference's contract (sorted lists in the binning chunk) holds. */
#define GSR_FLAG_NO_SORTED_LISTS 0x40u
/* Slow tiles first. A frame's blend lasts as long as its slowest tiles, and a camera moves little between two frames:
 * every tile's wave leaves how long it ran in a TILE HISTORY, and the next call that is given the same history sorts its
 * blend workgroups by it beside the depth sort, on a stream of the library's (blend 0.61 -> 0.53 ms from (0,0,-30) on the
 * bench scene). Outputs do not depend on it — the same tiles are composited the same way, sooner or later.
 * Which history a call takes: gsr_forward_args.tile_history, a gsr_tile_history the caller created (one per VIEW: a
 * rasterizer object, an eye of a stereo pair) — or, when that is NULL, one the library keeps per host thread, device and
 * STREAM of the call (at most eight streams per thread and device; calls on further streams run without). A history
 * belongs to the stream of the call that used it last: kernels of two calls that share a history are ordered by that
 * stream alone, so calls on different streams never share a default history (a ninth stream takes over the history the
 * thread has not used for the longest time). A caller who moves its OWN history to another stream: the library orders the
 * new stream behind the old one's work by an event it records on the OLD stream at that moment — which must therefore
 * still exist; a caller that has destroyed it (after waiting for it) says so first: gsr_tile_history_forget_stream. A history whose frames stop resembling each other — a trainer that draws an
 * unrelated camera every call, two views alternating on one history — is noticed (the two last frames' tile times are
 * compared when the order is sorted: each tile's SHARE of its frame's total tile time, the smaller of its two shares summed
 * over the tiles, below 0.8) and the order is dropped, patch order as without a history, until they do again: plan_used
 * then carries GSR_PLAN_TILE_ORDER_DROPPED. The order is sorted when the statistics of the calls before say the frame ends
 * on a few slow tiles or is a light one, on every call while the order is dropped (the sort is what looks whether the
 * frames resemble each other again), and every fourth call otherwise, to keep the statistics fresh. A history also carries the statistics by which the library decides, per call, whether the
 * blend runs beside the emission (GSR_FLAG_OVERLAP_EMIT). This flag switches all of it off for the call: it neither reads
 * nor writes a history and uses no second stream for it (GSR_TILE_HISTORY=0 in the environment does the same for every
 * call of the process). Frames of more than 32 768 tiles (beyond 3840 x 2160) keep no tile times either: the order is
 * sorted by one workgroup in LDS. */
#define GSR_FLAG_NO_TILE_HISTORY 0x80u
/* Deep tiles (csrc/blend.hip). A blend fed from the sorted lists may give a tile a workgroup of FOUR waves: they walk the
 * tile's list together (one fetch, one footprint test per entry) and composite one 16 x 4 strip each — a wave alone on its
 * SIMD issues a vector instruction every five cycles where the SIMD takes one every two or three, and a frame of small
 * splats lasts as long as the lone waves of its few deep tiles. Same pixels, finalT, nContrib, records_staged: every pixel
 * sees the same records in the same order. By default every tile of a frame with fewer than 16 instances per visible
 * Gaussian is composited that way (and every tile of a call with few tiles: a rank's band of a sharded frame); plan_used
 * then carries GSR_PLAN_DEEP_TILES. GSR_FLAG_NO_DEEP_TILES: never; GSR_FLAG_DEEP_TILES_ALL: whatever the frame (a
 * diagnostic, and what the tests compare the ordinary way against). Neither reads nor needs a tile history. */
#define GSR_FLAG_NO_DEEP_TILES 0x200u
#define GSR_FLAG_DEEP_TILES_ALL 0x400u
/* With deep tiles, EIGHT or SIXTEEN waves per tile (strips of 16 x 2 or 16 x 1 pixels; the upper lanes of a wave idle) instead
 * of four: what the library picks by itself when the view's tile history says the frame's work sits in a few hundred tiles
 * (a far view of a dense scene: a chip with one wave per SIMD issues at 40 % of what two waves per SIMD do). Diagnostics,
 * like GSR_FLAG_DEEP_TILES_ALL, which they imply; same outputs bit for bit. */
#define GSR_FLAG_DEEP_WAVES_8 0x800u
#define GSR_FLAG_DEEP_WAVES_16 0x1000u
enum { GSR_PLAN_SORT = 1, GSR_PLAN_BLOCKS = 2, GSR_PLAN_GENERIC = 3 /* grids wider than 255 tiles */,
       GSR_PLAN_LISTS_SKIPPED = 0x100 /* or-ed in: GSR_FLAG_NO_SORTED_LISTS took effect */,
       GSR_PLAN_BLEND_FROM_LISTS = 0x200 /* or-ed in: block plan whose blend read the sorted lists (sparse frames: fewer
                                            than 48 instances per visible Gaussian), not the block lists */,
       GSR_PLAN_TILES_REORDERED = 0x400 /* or-ed in: the blend started the slow tiles of the call before first
                                           (see GSR_FLAG_NO_TILE_HISTORY; informational) */,
       GSR_PLAN_EMIT_OVERLAPPED = 0x800 /* or-ed in: the blend ran beside the emission (see GSR_FLAG_OVERLAP_EMIT) */,
       GSR_PLAN_COLORS_BESIDE = 0x1000 /* or-ed in: geomState.rgb was written beside the scan / depth sort (or the blend), not by the preprocess */,
       GSR_PLAN_TILE_ORDER_DROPPED = 0x2000 /* or-ed in: the history's last frames did not resemble each other (another view
                                               every call): the blend took the patch order (informational) */,
       GSR_PLAN_DEEP_TILES = 0x4000 /* or-ed in: the blend composited its tiles by four waves each (see GSR_FLAG_NO_DEEP_TILES;
                                       informational) */ };

/* A tile history (see GSR_FLAG_NO_TILE_HISTORY): opaque, created for the CURRENT device, owned by the caller, one per view.
 * gsr_tile_history_destroy: the streams it was used on must be idle. */
typedef struct gsr_tile_history gsr_tile_history;
int gsr_tile_history_create(gsr_tile_history** out);
int gsr_tile_history_destroy(gsr_tile_history* history);
/* What the library last learnt about the history (host side, no device access; for tools and tests): out[0] = mean tile
 * time of its last sorted frame in units of 10 ns (0: none yet), [1] = that frame's longest tile (both as ONE wave would take:
 * a tile composited by four waves counts 2.5 times what it took), [2] = similarity of its two
 * last frames x 1000 (the smaller of a tile's two shares of its frame's tile time, summed over the tiles), [3] = 1 if the order is dropped at present,
 * [4] = calls since the history was last cleared (a new size), [5] = 1 if the last block-plan call ran its blend beside the
 * emission. */
int gsr_tile_history_stats(const gsr_tile_history* history, uint32_t out[6]);

/* The stream the history's last call ran on is gone (the caller waited for it and destroyed it): the next call with this
 * history, on whatever stream, is not ordered behind it. Host side only. */
int gsr_tile_history_forget_stream(gsr_tile_history* history);

/* For tools and tests: the tile times the history's LAST call recorded, times[tile] for the first `count` tiles of the frame
 * (row-major tile index) in units of 10 ns — bit 31 set: the tile was composited by four waves (a deep tile) —, and in
 * *deep_tiles (optional) how many leading entries of the last sorted order were deep tiles. Synchronises the device. */
int gsr_tile_history_times(const gsr_tile_history* history, uint32_t* times, int count, uint32_t* deep_tiles);

/* Gives back everything the library has allocated for the CALLING host thread, on every device (see the list at the top):
 * the second stream is drained first, so nothing of the library's own is in flight afterwards; work the thread's calls put
 * on the CALLER's streams is the caller's to wait for, as ever, before it frees the chunks. Receipts of the thread's
 * earlier calls can no longer be polled (gsr_poll_async_error reads their pinned words: do not pass one afterwards), and a
 * gsr_backward call for one of them runs without the forward's tile order. A later call of the thread starts from nothing
 * again. Also run when the thread ends. Always GSR_OK. */
int gsr_thread_release(void);

/* The environment switches (GSR_TILE_HISTORY, GSR_COLORS_BESIDE, GSR_COLORS_EARLY_PCT, GSR_FUSED_DEPTH, GSR_DEPTH_RECORDS, GSR_DEEP_ALL_MAX, GSR_BLOCK_FEED_MIN,
 * GSR_DEEP_WAVES_AUTO, GSR_DEEP_BY_HISTORY: A/B runs and tests, none needed in production) are read once per process, by the
 * first call that needs them; a test that changes one afterwards calls this to have them read again (no call in flight on
 * another thread meanwhile). */
void gsr_reread_environment(void);

/* The figures the launch heuristics derive from a device's compute-unit count (hipDeviceAttributeMultiprocessorCount, read
 * once per device — an MI355X in a partitioned mode shows fewer): out[0] = CUs, [1] = wave slots of the blend kernels
 * (4 SIMDs x 5 waves per CU), [2] = the same beside the emission's persistent workgroups (3 per SIMD), [3] = [1] again as
 * the divisor of the light-frame rule (250 us of tile time per slot). Pure: no device needed (tests/test_capi_cpu.py). */
void gsr_device_shape(int compute_units, uint32_t out[4]);

/* Receipt of one gsr_forward call: everything a LATER call (gsr_backward, gsr_poll_async_error) needs to know about it.
 * The reference keeps all per-call state in the caller-owned chunks (GSCuda.cu:723-725,734-736,782-784); so does this
 * library — the receipt only says which chunks those were, how many instances (R) and visible Gaussians (V) the call
 * found and which binning plan left which lists there. It is plain data: copy it, hand it to another host thread, keep
 * several (one per rasterizer). It describes the chunks as that call left them, so it is valid until the next
 * gsr_forward call that is given the same chunks. */
#define GSR_RECEIPT_MAGIC 0x31525347u   /* "GSR1" */
/* First word of BinningState.values after a call that left the sorted lists unwritten (GSR_FLAG_NO_SORTED_LISTS took
 * effect): no Gaussian index, so a gsr_backward call WITHOUT a receipt can tell that there is no list to walk. */
#define GSR_LISTS_SKIPPED_STAMP 0xFFFFFFFFu
typedef struct gsr_forward_receipt {
    uint32_t magic;                /* GSR_RECEIPT_MAGIC once gsr_forward has returned GSR_OK, else 0 */
    uint32_t plan_used;            /* as gsr_forward_args.plan_used */
    int32_t  num_gaussians, width, height;
    int32_t  tile_row_begin, tile_row_end;   /* the tile rows the call processed (0, ceil(H/16) for the whole frame) */
    uint32_t num_rendered;         /* R */
    uint32_t num_visible;          /* V: Gaussians with at least one tile in this call */
    uint32_t serial;               /* which call of its host thread and device */
    char*    geometry_chunk;       /* what the three allocator callbacks returned (binning: NULL if R == 0) */
    char*    image_chunk;
    char*    binning_chunk;
    const volatile uint32_t* async_words;   /* host memory (pinned, never freed): {N-sized sort gave up, R-sized sort gave up
                                               (each: 0, or the serial of the call whose kernel gave up), serial of the call
                                               that owns the words, 0} — see gsr_poll_async_error */
    gsr_tile_history* tile_history;         /* gsr_forward_args.tile_history of the call (NULL: one of the library's own). A caller's
                                               history must outlive the receipts that name it (gsr_backward looks into it for
                                               the forward blend's tile order) */
} gsr_forward_receipt;

/* Arguments of one forward call. Fields up to box_max are, in order, the parameters of
 * gscuda::forward (GSCuda.cuh:103-126); the rest are extensions with neutral defaults (0). */
typedef struct gsr_forward_args {
    uint32_t struct_size;          /* = sizeof(gsr_forward_args) */
    uint32_t flags;
    gsr_alloc_fn geometry_alloc;   void* geometry_user;
    gsr_alloc_fn binning_alloc;    void* binning_user;
    gsr_alloc_fn image_alloc;      void* image_user;
    int32_t num_gaussians;
    int32_t sh_dims;               /* D, unused by the gscuda semantics */
    int32_t M;                     /* unused */
    const float* background;       /* device vec3 */
    int32_t width, height;
    const float* means3D;          /* device vec4[N] */
    const float* shs;              /* device f32[48 N], DC first */
    const float* colors_precomp;   /* device vec3[N] or NULL */
    const float* opacities;        /* device f32[N] */
    const float* scales;           /* device vec4[N] */
    float scale_modifier;
    const float* rotations;        /* device vec4[N], real part first */
    const float* cov3D_precomp;    /* device f32[6 N] or NULL */
    const float* view_matrix;      /* device 16 f32 column-major, row 2 negated */
    const float* proj_matrix;      /* device 16 f32 column-major, perspective*view */
    const float* cam_pos;          /* device vec3, unused */
    float tan_fovx, tan_fovy;
    int32_t prefiltered;           /* unused */
    float* out_color;              /* device f32[3 W H], planar CHW */
    int32_t* radii;                /* device i32[N] or NULL (falls back to internal_radii) */
    int32_t* rects;                /* device i32[2 N] or NULL (radius-based rectangles) */
    const float* box_min;          /* unused (reference overrides with -FLT_MAX) */
    const float* box_max;          /* unused */
    /* ---- extensions ---- */
    void* stream;                  /* hipStream_t; NULL = the default stream */
    int32_t tile_row_begin;        /* multi-GPU: only tile rows [begin,end) are binned,   */
    int32_t tile_row_end;          /*   sorted and blended; 0,0 = all rows. In such a call  */
                                   /*   a Gaussian without a tile in the band counts as     */
                                   /*   invisible (radius 0, its geometry fields unwritten) */
    gsr_tile_history* tile_history;/* the view's tile history, or NULL: the library's own for this thread, device and
                                      stream (see GSR_FLAG_NO_TILE_HISTORY) */
    /* ---- outputs ---- */
    uint32_t num_rendered;         /* R = sum of tiles touched (for the rows processed)    */
    uint64_t records_staged;       /* R_f, only with GSR_FLAG_COUNT_STAGED                 */
    float stage_ms[GSR_NUM_STAGES];/* only with GSR_FLAG_PROFILE                           */
    uint32_t plan_used;            /* GSR_PLAN_* of this call (0 if R == 0), | GSR_PLAN_LISTS_SKIPPED */
    gsr_forward_receipt receipt;   /* hand it to gsr_backward / gsr_poll_async_error (magic = 0 unless GSR_OK is returned) */
} gsr_forward_args;

/* The forward pass. Calls geometry_alloc(required_geometry(N)), then
 * image_alloc(required_image(W*H)+128), then — only if R>0, after the scan —
 * binning_alloc(required_binning(R)+128): once each, in that order (GSCuda.cu:723-784).
 * Synchronises `stream` once mid-call to read R back (GSCuda.cu:772). Returns GSR_*.
 * R == 0 returns GSR_OK and leaves out_color untouched (GSCuda.cu:775-778). */
int gsr_forward(gsr_forward_args* args);

/* Sticky last error of the calling thread's most recent gsr_* call + its text. */
int gsr_last_error(void);
const char* gsr_error_string(int code);
/* Text of the HIP error behind the last GSR_ERR_HIP ("" if none). */
const char* gsr_last_hip_error(void);

/* After the caller has synchronised the stream of the gsr_forward call that issued `receipt`: GSR_OK, or
 * GSR_ERR_INTERNAL if a radix-sort look-back wait expired during that call (the frame is then invalid). Mirrors the
 * reference caller polling the sticky error after its sync (CudaBuffer.hpp:8-12). Any thread may ask. The words a call
 * reports into are one of 64 slots per host thread and device, handed out in turn: asked about a call that lies more
 * than 63 calls back the answer is GSR_ERR_STALE_RECEIPT (its slot has a new owner; nothing is known any more). A kernel
 * that gives up writes its own call's serial, so a call still running on another stream when its slot comes round again
 * cannot raise the new owner's flag. NULL or a receipt without the magic: GSR_ERR_INVALID_ARG. */
int gsr_poll_async_error(const gsr_forward_receipt* receipt);

/* getHigherMsb (GSCuda.cu:481-502): bits of the tile id that take part in the sort. */
uint32_t gsr_higher_msb(uint32_t n);

/* ---- stage entry points (unit-level parity tests; all asynchronous on `stream`) ---- */
/* Inclusive u32 prefix sum (replaces cub::DeviceScan::InclusiveSum, GSCuda.cu:771).
 * temp needs gsr_scan_temp_bytes(n) bytes. in == out is allowed. */
size_t gsr_scan_temp_bytes(size_t n);
int gsr_inclusive_scan_u32(const uint32_t* in, uint32_t* out, size_t n, char* temp, void* stream);
/* Stable LSD radix sort of (u64 key, u32 value) pairs on key bits [begin_bit, end_bit)
 * (replaces cub::DeviceRadixSort::SortPairs, GSCuda.cu:794-797). Inputs are preserved. */
size_t gsr_sort_temp_bytes(size_t n);
int gsr_sort_pairs_u64_u32(const uint64_t* keys_in, uint64_t* keys_out, const uint32_t* values_in,
                           uint32_t* values_out, size_t n, int begin_bit, int end_bit, char* temp, void* stream);

/* The colours of the reference's semantics, once per scene: colors[3 i + c] = 0.5f + 0.4f * shs[48 i + c] (GSCuda.cu:362-366;
 * the colour does not depend on the view). Passed as `colors_precomp` (GSCuda.cuh:111; the blend then reads them instead of
 * geomState.rgb, GSCuda.cu:803) they spare every frame the 12-byte read at a 192-byte stride — one 128-byte line per visible
 * Gaussian — and the 12 N bytes of geomState.rgb: same pixels bit for bit. Device pointers; asynchronous on stream. */
int gsr_colors_from_dc(int n, const float* shs, float* colors, void* stream);

/* The blend's exponential (csrc/blend_core.hpp, exp_ref: the float exponential of GSCuda.cu:645 computed as glibc's expf
 * computes it), argument by argument: out[i] = exp(in[i]), in[i] <= 0.5 and not NaN; device pointers. For the tests: it must
 * agree bit for bit with the host's expf, which is what makes alpha, T and every threshold decision of the blend those of the CPU oracle. */
int gsr_blend_expf(int n, const float* in, float* out, void* stream);

/* The blend's footprint test (csrc/blend_core.hpp), record by record: misses[i] = 1 iff the library would drop record i
 * (centre means2D[i], conic + opacity conic_opacity[i]) when it stages the list of tile tile_xy[i] = (tx, ty) of a
 * width x height image, i.e. iff it has PROVEN that no pixel of that tile passes alpha >= 1/255 (GSCuda.cu:645-646).
 * All pointers are device pointers. For the tests: a dropped record that could light a pixel would be a parity bug. */
int gsr_footprint_misses_tile(int n, const float* means2D, const float* conic_opacity, const int32_t* tile_xy,
                              int width, int height, uint8_t* misses, void* stream);

/* ---- backward pass (next row after the hot path: BASELINE config 5) ----
 * Gradients of L = sum(dL_dout_color * out_color) of ONE gsr_forward call (gscuda semantics) w.r.t. the
 * per-Gaussian quantities its blend loop reads, and on to the 3-D covariance and the DC harmonics. The
 * reference has no backward pass (parity unpinned); conventions, checked by finite differences in
 * oracle/backward_np.py:
 *   dL_dmean2D        vec2[N]  w.r.t. the pixel-space centre (no NDC factor)
 *   dL_dconic_opacity vec4[N]  w.r.t. (A, B, C) of power = -0.5 (A dx^2 + C dy^2) - B dx dy, and opacity
 *   dL_dcolors        vec3[N]
 *   dL_dcov2D         vec4[N]  w.r.t. the 2-D covariance, (m00, m01, m11, 0) (optional; what the chain below should start from)
 *   dL_dcov3D         f32[6N]  w.r.t. the six stored covariance numbers (optional; without sums_f64, NULL skips the chain)
 *   dL_dshs           f32[48N] colour = 0.5 + 0.4 DC: the DC triple of every Gaussian, = 0.4 dL_dcolors, and zeros in the 13
 *                              floats behind it (one whole 64-byte write per Gaussian; floats 16..47 are not touched);
 *                              optional — a caller that scales dL_dcolors itself saves 0.18 ms of strided writes per frame
 *   dL_dmeans3D       vec4[N]  (x, y, z, 0) through the pixel-space centre and through the Jacobian of cov2D; optional
 *   dL_dscales        vec4[N]  (x, y, z, 0);  dL_drotations vec4[N] w.r.t. the quaternion as given; optional
 *                              (not available when the forward call took cov3D_precomp)
 * Every output is optional once sums_f64 is given (the sums then live there and the float arrays are only copies): a NULL
 * output is not computed where that saves work and never written. The chain writes 150-odd bytes per Gaussian when all are
 * asked for; BASELINE config 5's set (dL_dmean2D, dL_dcov3D, dL_dshs) is 96 of them. Without sums_f64 the three arrays of the
 * sums are required, and dL_dmeans3D / dL_dscales / dL_drotations need dL_dcov3D.
 * Hard tests of the forward (power > 0, alpha < 1/255, transmittance cut-off) select a branch; where
 * alpha is clamped to 0.99 its derivative w.r.t. the Gaussian's parameters is zero. The state pointers are
 * those of the forward call's chunks (gsr_*_from_chunk): it must have run on the same inputs, same size.
 * With flags = GSR_FLAG_SEMANTICS_INRIA (the forward call's flag) the chain is the upstream profile's: two focal
 * lengths, w epsilon 1e-7, the raw quaternion, and colour = max(0, 0.5 + SH(view direction)) — dL_dshs then receives
 * all (sh_dims + 1)^2 coefficient triples of every Gaussian (zeros beyond them and for culled Gaussians; a channel
 * that was clamped at zero passes nothing), and dL_dmeans3D the term through the view direction. */
typedef struct gsr_backward_args {
    uint32_t struct_size;          /* = sizeof(gsr_backward_args) */
    uint32_t flags;                /* GSR_FLAG_PROFILE */
    int32_t num_gaussians, width, height;
    const float* background;       /* device vec3 */
    /* state of the forward call */
    const float* means2D;          /* geometry chunk */
    const float* conic_opacity;
    const float* colors;           /* geometry rgb, or the colors_precomp the forward call was given */
    const float* cov3D;            /* geometry cov3D, or the cov3D_precomp given (only for the chain) */
    const int32_t* radii;          /* internal_radii or the radii buffer given (only for the chain) */
    const uint32_t* ranges;        /* image chunk */
    const uint32_t* n_contrib;
    const float* final_t;          /* accum_alpha */
    const uint32_t* point_list;    /* binning chunk: values (with a receipt: must be the values array of ITS binning chunk) */
    /* inputs of the forward call that the covariance chain needs again */
    const float* means3D;
    const float* view_matrix;
    float tan_fovx, tan_fovy;
    const float* dL_dout_color;    /* device f32[3 W H], planar like out_color */
    /* outputs (device) */
    float* dL_dmean2D;             /* these three: required without sums_f64, optional with it */
    float* dL_dconic_opacity;
    float* dL_dcolors;
    float* dL_dcov3D;              /* or NULL */
    float* dL_dshs;                /* or NULL */
    float* dL_dcov2D;              /* vec4[N] (m00, m01, m11, 0): w.r.t. the full symmetric 2-D covariance (a, b; b, c) of
                                      GSCuda.cu:226-230, summed pixel by pixel by the render backward; or NULL. The chain to
                                      dL_dcov3D and beyond starts from it when given — without it the chain derives it as
                                      -K (dL/dK) K from dL_dconic_opacity, which loses cond(K)^2 of the sums' digits (a splat
                                      that fills the screen: 1e6) */
    double* sums_f64;              /* f64[12 N] scratch or NULL. ZERO on entry, left zero on return. With it the twelve sums of a
                                      Gaussian (the four arrays above) are accumulated in double — a splat that fills the screen
                                      collects terms of either sign from 8 160 tiles, and in float their order of arrival shows
                                      in the fourth digit of its gradients, differently every run — and rounded to float once;
                                      the float arrays need not be cleared by anybody then */
    /* chain down to the inputs (all optional; without sums_f64 they need dL_dcov3D) */
    const float* proj_matrix;      /* inputs of the forward call */
    const float* scales;
    const float* rotations;
    float scale_modifier;
    float* dL_dmeans3D;            /* or NULL */
    float* dL_dscales;             /* or NULL */
    float* dL_drotations;          /* or NULL (needs dL_dscales) */
    void* stream;
    int32_t tile_row_begin, tile_row_end;   /* as gsr_forward: the rows the forward call processed */
    float stage_ms[2];             /* with GSR_FLAG_PROFILE: render backward, covariance / colour chain */
    /* ---- GSR_FLAG_SEMANTICS_INRIA: the forward call ran with the upstream semantics ---- */
    const float* cam_pos;          /* inputs of the forward call (needed with dL_dshs) */
    const float* shs;              /* device f32[48 N], [16][3] per Gaussian */
    const uint8_t* clamped;        /* geometry chunk: bool[3 N], colour channel was clamped at zero */
    int32_t sh_dims;               /* SH degree the forward call evaluated (0..3) */
    /* ---- which forward call this is the backward of ---- */
    gsr_forward_receipt receipt;   /* gsr_forward_args.receipt of that call, by value. With it the block lists that call left
                                      in its chunks are used where it ran the block plan (all tiles if GSR_FLAG_NO_SORTED_LISTS
                                      left `values` unwritten, else the tiles of shallow blocks, keysUnsorted serving as
                                      scratch for per-entry sums), and R == 0 gives all-zero gradients. All zero (no receipt):
                                      the reference's contract only — the sorted lists must be in point_list; the call then
                                      reads point_list[0] back (one stream sync) and refuses GSR_LISTS_SKIPPED_STAMP with
                                      GSR_ERR_INVALID_ARG. A receipt that does not fit the other arguments (sizes, rows,
                                      point_list): GSR_ERR_INVALID_ARG. */
} gsr_backward_args;
int gsr_backward(gsr_backward_args* args);

/* ---- point-splat path (gscuda::forwardPoints, GSCuda.cuh:19-42 / GSCuda.cu:26-155; the reference never
 * calls it) ---- Same argument struct as gsr_forward; read: num_gaussians, width, height, background,
 * means3D (a stride of THREE floats here, GSCuda.cu:65), shs, proj_matrix, out_color, geometry_alloc (asked for
 * 32 bytes, as the reference does for its empty pc::GeometryState), image_alloc, stream. Every centre inside
 * NDC [-1,1]^2 x [0,1] lands on one pixel; the nearest wins (the lowest index among equal depths — the
 * reference's unordered colour write makes its own result timing dependent). */
typedef struct gsr_points_image_state {   /* pc::ImageState (AuxBuffer.cuh:26-33), `size` = width*height */
    float*    depth;           /* f32[size] NDC z of the winner, 1.0 where none   */
    float*    out_color;       /* f32[3 size] planar temporary image              */
    float*    default_depth;   /* f32[1] = 1.0                                    */
    uint64_t* winner;          /* u64[size] (depth bits << 32 | index), this library's depth-test table */
} gsr_points_image_state;
char* gsr_points_image_from_chunk(char* chunk, int size, gsr_points_image_state* out);
size_t gsr_required_points_image(int size);
int gsr_forward_points(gsr_forward_args* args);

/* ---- multi-GPU: the row-band exchange of a frame sharded by screen tile rows (the reference is single-GPU:
 * nothing of this replaces a reference interface) ----
 * One process per GPU. Every rank holds the whole planar frame (3 x height x width floats, device memory) and has rendered
 * tile rows [bounds[rank], bounds[rank + 1]) of it (gsr_forward_args.tile_row_begin / _end); one call moves every band
 * into place in the peers' frames: ncclGroupStart, one ncclSend / ncclRecv per (peer, colour plane) of exact size, ncclGroupEnd
 * — no staging, no padding. RCCL is loaded at run time from `rccl_path` (NULL: librccl.so.1; a process that uses PyTorch
 * passes the copy torch has mapped), so the library does not depend on it.
 *   gsr_exchange_unique_id : one rank creates the 128-byte id (ncclGetUniqueId) and hands it to the others by any means
 *   gsr_exchange_create    : every rank, with ITS device current (ncclCommInitRank: collective, blocks until all have called)
 *   gsr_exchange_bands     : bounds = world + 1 tile-row boundaries (0 ... ceil(height / 16), non-decreasing; empty bands
 *                            allowed); root < 0: every rank receives every band (all-gather); root = r: only rank r
 *                            receives (gather: 1 / world of the traffic when one rank displays). Asynchronous on `stream`.
 *   gsr_exchange_plan      : host only, no communicator: the transfers a rank would issue, in order (tests, tools)
 *   gsr_exchange_loopback  : `planes` pieces of `count` floats sent from src to dst on the OWN rank through the same
 *                            group of ncclSend / ncclRecv (a diagnostic: what a single GPU can run of the transfers)
 * All return GSR_OK or an error code (gsr_exchange_plan: the count, or the negated code); gsr_exchange_last_error() has
 * the RCCL / loader message behind a GSR_ERR_HIP. */
typedef struct gsr_exchange gsr_exchange;
int gsr_exchange_unique_id(const char* rccl_path, char* id128);
int gsr_exchange_create(const char* rccl_path, const char* id128, int rank, int world, gsr_exchange** out);
int gsr_exchange_bands(gsr_exchange* x, float* frame, int width, int height, const int32_t* bounds, int root, void* stream);
int gsr_exchange_plan(int rank, int world, int width, int height, const int32_t* bounds, int root, int max_ops,
                      int32_t* is_send, int32_t* peer, uint64_t* offset, uint64_t* count);
int gsr_exchange_loopback(gsr_exchange* x, const float* src, float* dst, uint64_t count, int planes, void* stream);
int gsr_exchange_destroy(gsr_exchange* x);
const char* gsr_exchange_last_error(void);

/* ---- scene loading (next row after the hot path) ---- */
/* Header of a 3DGS .ply as the reference reads it (apps/gsrast/SplatData.cpp:114-145): vertex
 * count = third token of the third line; data starts after the "end_header" line. Host only.
 * Returns GSR_ERR_INVALID_ARG if the file cannot be opened or has no end_header. */
int gsr_ply_parse_header(const char* path, int* num_splats, long long* data_offset);
/* Activations of SplatData::loadFromPly (SplatData.cpp:28-66) on the GPU: raw_device holds n
 * records of 62 floats (SplatData.hpp:17-25) in HBM; outputs are the SoA gsr_forward takes
 * (means3D/scales/rotations vec4[n], opacities f32[n], shs f32[48 n]). Asynchronous on stream. */
int gsr_ply_activate(const float* raw_device, int n, float* means3D, float* scales, float* rotations,
                     float* opacities, float* shs, void* stream);
/* The 48 SH floats of a record are f_dc_0..2 followed by f_rest_0..44, and a 3DGS .ply stores f_rest CHANNEL-major
 * (R1..R15, G1..G15, B1..B15). The reference copies them as they lie (SplatData.cpp:146) and reads only the DC
 * triple (GSCuda.cu:362-366): GSR_SH_LAYOUT_FILE, what gsr_ply_activate does. GSR_FLAG_SEMANTICS_INRIA reads shs as
 * [N][16][3], coefficient-major (upstream's layout, SURVEY.md divergence D2): a scene that is to be drawn with that
 * flag and sh_dims >= 1 must be loaded with GSR_SH_LAYOUT_COEFFICIENT_MAJOR, which transposes f_rest on the way. */
enum { GSR_SH_LAYOUT_FILE = 0, GSR_SH_LAYOUT_COEFFICIENT_MAJOR = 1 };
int gsr_ply_activate_layout(const float* raw_device, int n, float* means3D, float* scales, float* rotations,
                            float* opacities, float* shs, int sh_layout, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GSRAST_AMD_H */
