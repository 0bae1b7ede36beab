// Tables shared between the host planner (pqd_api.cpp) and the kernels (pqd_kernels.hip).
// All of them live in HBM; the host builds the static ones once per plan.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pqd {

typedef uint8_t u8;
typedef uint32_t u32;
typedef uint64_t u64;

constexpr u32 ERR_POS = 0xFFFFFFFFu;
constexpr u32 UNKNOWN_POS = 0xFFFFFFFEu;  // group exit not tabulated
constexpr u32 NO_SLOT = 0xFFFFFFFFu;      // an entry without a row in the exit table

// device status codes (mapped to pqd_status by the host)
constexpr int ST_CORRUPT = 1, ST_NONCANONICAL = 2, ST_UNSUPPORTED = 3, ST_INTERNAL = 4;  // INTERNAL: a workgroup gave up waiting for another

// parquet physical types
constexpr int PT_BOOLEAN = 0, PT_BYTE_ARRAY = 6;

// ---- PLAIN value gather -------------------------------------------------------------------
// One wave per workgroup, 16 bytes per lane.  Measured on the batch of all column kinds (the gather beside the level and
// string kernels) and on the gather alone: 64 x 1: 3.48 ms / 6.00 TB/s; 256 x 1: 3.60 / 6.03; 128 x 2: 3.62 / 5.95;
// 256 x 2: 3.54 / 5.86; 64 x 4: 3.84 / 5.45.  Workgroups of one wave fill single wave slots as the four-wave workgroups
// of the other kernels leave them.
#ifndef CP_THREADS_N
#define CP_THREADS_N 64
#endif
#ifndef CP_UNROLL_N
#define CP_UNROLL_N 1
#endif
constexpr int CP_THREADS = CP_THREADS_N;
constexpr int CP_UNROLL = CP_UNROLL_N;
constexpr u32 CP_BLK = CP_THREADS * 16 * CP_UNROLL;  // bytes per block (1 KiB): many small blocks beat a grid-stride loop here
constexpr u32 CP_GRP = 256;  // the planner tabulates the segment of every CP_GRP-th block: a block finds its own from there in a step or two

struct Seg {
    u64 src_off;  // byte offset in the source buffer
    u8* dst;      // destination address
    u64 len;
};

// ---- chain resolution (run headers, BYTE_ARRAY lengths) ---------------------------------------
constexpr int KIND_LEVELS = 0, KIND_STRINGS = 1, KIND_SNAPPY = 2;
constexpr u32 CH_SUB = 32;        // source bytes per lane (sub-tile)
constexpr u32 CH_THREADS = 64;    // one wavefront per tile
constexpr u32 CH_T = CH_THREADS * CH_SUB;  // source bytes per tile (2048)
constexpr u32 CH_WIN = CH_SUB / 4 + 2;    // dwords a lane holds: its sub-tile + 8 bytes of lookahead
constexpr u32 CH_E_LEVELS = 256;  // tabulated entry positions per tile: covers the reference writer's longest
                                  // run (1 + 63 groups x 4 bytes); longer jumps take the serial hop
constexpr u32 CH_E_STRINGS = 8;   // rows per tile: its first plausible string starts (see StrIndex)

constexpr u32 CH_SERIAL = 32;     // level tiles with at most this many runs are marked by a serial walk
constexpr u32 CH_G_LEVELS = 8;    // tiles per hop group (level blocks are ~60 tiles per 1M-value page)
constexpr u32 CH_G_STRINGS = 16;  // tiles per hop group (string pages are thousands of tiles: they get a super level)
constexpr u32 CH_G_SNAPPY = 64;   // tiles per hop group of a snappy block
constexpr u32 CH_SUPER = 16;      // groups per super-group (strings only)
constexpr u32 chain_group_tiles(int kind) { return kind == KIND_LEVELS ? CH_G_LEVELS : kind == KIND_STRINGS ? CH_G_STRINGS : CH_G_SNAPPY; }

constexpr u32 OT = 4096;          // level values per expand tile
constexpr unsigned long long SN_WAIT_TICKS = 200000000ull;  // snappy: a workgroup waits this long (100 MHz wall clock: 2 s) for an earlier tile's flag, then gives up
constexpr u32 SN_T = 32768;       // snappy: output bytes per decode tile (half the block size of the usual encoders: 15-bit tile positions)
constexpr u32 EX_CAP = 1024;      // runs staged in LDS per expand batch
constexpr u32 BT = 4096;          // bools per tile
constexpr u32 RT = 4096;          // levels per row-offset tile

struct Region {
    u64 src_off;     // start of the chain bytes (after the 4-byte length for level blocks)
    u32 len;         // chain bytes
    u32 width;       // level bit width
    u32 tile_base;   // first tile of the region in the global tile numbering
    u32 otile_base;  // first expand tile (levels)
    u64 run_base;    // first entry of the region in the run table (levels); page path: level index of the page's first level in its chunk
    u64 n_out;       // levels: values to emit (page num_values); strings: upper bound of values in the page
    u8* out;         // levels: destination of value 0; strings: the chunk's offsets array
    u8* out2;        // strings: the chunk's chars array; page path, definition levels: the chunk's validity bitmap (zeroed before the run)
    u32 page;        // page index
    u32 count_level; // levels: count values equal to this into page_dyn[page].nn; ERR_POS = do not count
    u32 group_base;  // first hop group of the region in the global group numbering
    u32 lv_base;     // levels, single-pass decode: first super-tile of the region in the granule numbering
};

struct ExitRec { u32 tgt; u32 cnt; u64 vals; };
struct StrExit { u32 tgt; u32 cnt; };  // the tile-level rows of string chains (no value counts); same array
struct TileEntry { u32 pos; u32 cnt; u64 vals; u32 n; u32 slot; };  // tile entries only: n = elements that start in the tile, slot = the entry's row of the exit table
struct RegionTotal { u32 cnt; u32 err; u64 vals; };
struct RunRec { u32 out_off; u32 info; };

// ---- pages and chunks ----------------------------------------------------------------------
struct PageRec {
    u64 val_src_off;  // source offset of the values region (payload after the level blocks)
    u32 val_len;      // bytes in the values region
    u32 num_values;   // DataPageHeader.num_values (levels, nulls included)
    u32 chunk;
    u32 pad;
};
struct PageDyn {
    u32 nn;        // non-null values of the page (def == max_def)
    u32 pad;
    u64 val_off;   // dense value index of the page's first value within its chunk
    u64 char_off;  // strings: char index of the page's first byte within its chunk
};
struct ChunkRec {
    int phys_type, max_def, max_rep;
    u32 width_bytes;  // 4 / 8 for fixed-width types, else 0
    u32 first_page, n_pages;
    u64 n_levels;
    u64 stream_len;   // sum of the pages' values-region lengths
    u64 req_values;   // required chunk: parquet.Page.N, the number of values the typed Read takes
    u8 *defs, *reps, *values, *offsets, *chars, *validity, *page_nonnull, *row_level_off, *row_value_off, *value_index;
    u8* row_starts;   // repeated chunks: bit i = level i has repetition level 0 (written by the level decode, read by k_rows)
    u32 shares_rep;   // reps, row_starts and row_level_off are those of another chunk with byte-identical repetition level blocks: not written for this one
    u32 pad2;
};
struct ChunkDyn {
    u64 n_values, chars, n_rows;
};
struct RowTile { u32 rows, vals; u64 row_base, val_base; };

// ---- launchers -----------------------------------------------------------------------------
hipError_t launch_copy(hipStream_t st, const u8* src, const Seg* segs, const u32* blk_prefix, const u32* blk_grp, u32 n_segs,
                       u32 n_blocks, u32 max_grid);
// strings: which positions of a chain tile are plausible string starts (see entry_slot in pqd_kernels.hip)
struct StrIndex {
    const unsigned short* cand;  // n_tiles * CH_E_STRINGS: tile offsets of the first plausible positions, ascending, 0xFFFF = none
    const unsigned short* ncand; // n_tiles: plausible positions of the tile (counted until its rows were full)
    unsigned short* gpre;        // n_tiles: plausible positions in the earlier tiles of the tile's group (k_chain_groups)
    unsigned short* spre;        // n_tiles: ... of the tile's super-group (k_chain_supers)
};
// strings, single pass: SF_UNIT bytes of one page, decoded by one workgroup
struct StrUnit {
    u64 src_off;  // source offset of the unit's first byte
    u32 tb0;      // page position of that byte
    u32 rem;      // bytes from there to the end of the page
    u32 region;   // the page's string region
    u32 ui;       // index of the unit in page order within its chunk (its status word)
};
struct ChainArgs {
    const Region* regions;
    const u32* tile_prefix;   // n_regions + 1
    const u32* group_prefix;  // n_regions + 1
    u32 n_regions, n_tiles, n_groups;
    ExitRec* exit_tbl;        // n_tiles * E
    ExitRec* gexit;           // n_groups * E
    TileEntry* tile_entry;    // n_tiles   (reset to 0xFF before every run)
    TileEntry* group_entry;   // n_groups  (reset to 0xFF before every run)
    RegionTotal* totals;      // n_regions
    void* str_cp;             // strings: n_tiles * CH_E_STRINGS rows of 8 u16 walk checkpoints
    const u32* super_prefix;  // n_regions + 1; super-groups are optional (n_supers == 0: three levels)
    u32 n_supers;
    ExitRec* sexit;           // n_supers * E
    TileEntry* super_entry;   // n_supers  (reset to 0xFF before every run)
    const u32* tile_region;   // n_tiles: the region of every tile (optional; null: the kernels search tile_prefix)
    StrIndex str_index;       // strings only (written by pass 1)
    // strings, single pass (k_str_fused): null = the two passes decode every chunk
    u64* str_stat;            // n_units status words {state, strings, exit}; zero before every run
    u32 n_units;              // units of SF_UNIT bytes of one page
    const StrUnit* units;     // n_units, in launch order: the k-th unit of every chunk before the k+1-th of any
    u64* str_stamps;          // diagnostics only (else null): SF_STAMPS wall-clock stamps (100 MHz) per workgroup of k_str_fused
    u32* str_failed;          // one word per BYTE_ARRAY chunk, zero before every run: set = the chunk goes through the two passes
    u32* str_any;             // one word, zero before every run: set when any chunk was given up
    u32* str_repairs;         // one word per BYTE_ARRAY chunk, zero before every run: tiles of the chunk that were walked string by string
};
hipError_t launch_chain(hipStream_t st, int kind, const u8* src, const ChainArgs& a, RunRec* run_tbl, u32* otile_first_run,
                        const PageDyn* page_dyn, int* status, int phase);
// strings in one pass over the page bytes; the chunks it gives up are flagged in a.str_failed and left to
// launch_chain(KIND_STRINGS), whose kernels then skip every other chunk.  str_pd: one entry per string region.
// Region fields on this path: count_level = first unit of the region, lv_base = first unit of the region's chunk,
// run_base = bytes of the chunk's earlier pages (values regions), width = entries of the chunk's offsets array,
// otile_base = index of the chunk among the BYTE_ARRAY chunks.
#ifndef SF_WAVES_N
#define SF_WAVES_N 4
#endif
constexpr u32 SF_WAVES = SF_WAVES_N;           // waves per workgroup (unit)
constexpr u32 SF_SUB = 64, SF_T = 64 * SF_SUB; // bytes per lane and per tile of this kernel (a tile is one wave's)
constexpr u32 SF_TPU = SF_WAVES;               // tiles per unit
constexpr u32 SF_UNIT = SF_TPU * SF_T;         // bytes per unit
constexpr u32 SF_LDS_WORDS = SF_T / 4 + 4;     // a tile's image in LDS: its bytes, four bytes of lookahead, room for the staged offsets
constexpr u32 SF_REP_MAX = 2;                  // tiles of a unit that may be walked string by string
static_assert(SF_TPU <= 64, "one lane per tile of a unit");
constexpr u32 SF_STAMPS = 16;
// ff_begin / ff_bytes: the entry tables of the two passes (a multiple of 16 bytes), set to 0xFF here when a chunk was given up
hipError_t launch_str_fused(hipStream_t st, const u8* src, const ChainArgs& a, PageDyn* str_pd, u8* ff_begin, size_t ff_bytes);
// otile_region: region of every expand tile plus one spare entry (may be null: the kernel then searches
// otile_prefix); otile_first_run must have one spare entry after its n_otiles.
hipError_t launch_levels_expand(hipStream_t st, int esz, const u8* src, const Region* regions, const u32* otile_region,
                                const u32* otile_prefix, u32 n_regions, u32 n_otiles, const RunRec* run_tbl,
                                const u32* otile_first_run, const RegionTotal* totals, u32* otile_cnt, PageDyn* page_dyn);
// ---- single-pass level decode of the page path (pqd_levels.hip) ---------------------------------
// A super-tile is LV_T chain bytes of one level block and belongs to one workgroup, which resolves the run
// headers inside it, learns where the true chain enters from the super-tile before it (one 8-byte granule per
// boundary), and expands its runs straight into the level array.
// The kernel is built twice (pqd_levels.hip and, through pqd_levels_4k.hip, with PQD_LV_TILE_BYTES = 4096): 8 KiB
// super-tiles make the hand-off chain of a block short (16 hops per 1M-level page), which is what a small batch
// waits for; 4 KiB super-tiles (twice the workgroups, half the LDS each) run a large batch 10-20 % faster.
#ifndef PQD_LV_TILE_BYTES
#define PQD_LV_TILE_BYTES 8192
#endif
constexpr u32 LV_T = PQD_LV_TILE_BYTES;   // chain bytes per super-tile in this translation unit
constexpr u32 LV_THREADS = LV_T / 16;     // one 16-byte sub-tile per thread
constexpr u32 LV_T_SMALL_BATCH = 8192, LV_T_LARGE_BATCH = 4096;
// A batch whose 8 KiB super-tiles are more than this many times those of its longest block takes the 4 KiB build: below that the
// step waits for the hand-over chain of the longest block, which 8 KiB super-tiles halve.  Measured (4 KiB against 8 KiB, ms):
// optional int32, blocks of 16 super-tiles: 4 blocks 0.097 / 0.087, 16: 0.108 / 0.097, 32: 0.116 / 0.127, 64: 0.150 / 0.176;
// nested columns (longest block a fifth of a page's): ratio 3: 0.179 / 0.142, 12: 0.189 / 0.162, 49: 0.294 / 0.316;
// nine column kinds, ratio 25: 0.413 / 0.400, 50: 0.526 / 0.571; optional strings, 32 blocks: 0.551 / 0.574.
constexpr u32 LV_CHAIN_RATIO = 28;
constexpr u32 LV_SUB = LV_T / LV_THREADS;
constexpr unsigned long long LV_WAIT_TICKS = 200000000ull;  // a workgroup waits this long (100 MHz wall clock: 2 s) for a granule, then gives up
struct LvTile { u32 region, index; };  // launch order: super-tile `index` of `region`; predecessors come first
struct LvArgs {
    const Region* regions;
    const LvTile* order;   // n_tiles entries sorted by (index, region)
    u32 n_tiles;
    u64* pos_g;            // n_tiles granules {1 << 32 | exit position}; zero before every launch
    u64* val_g;            // n_tiles granules {1 << 32 | values up to the super-tile's end, saturated}; zero before every launch
    u32* ticket;           // one word, zero before every launch
    PageDyn* page_dyn;     // nn of the counted regions' pages is added to (zero before every launch); may be null
    int* status;
    u64* stamps;           // diagnostics only (else null): LV_STAMPS wall-clock stamps (100 MHz) per ticket
};
constexpr u32 LV_STAMPS = 16;
// tile_bytes = LV_T_SMALL_BATCH or LV_T_LARGE_BATCH: the super-tile size the tables of `a` were planned for
hipError_t launch_levels(hipStream_t st, const u8* src, const LvArgs& a, u32 tile_bytes);
hipError_t launch_levels_4k(hipStream_t st, const u8* src, const LvArgs& a);  // the second build
constexpr u32 LV_ALLOW_SHORT = 1;  // Region.otile_base flag on this path: a block with fewer values than n_out is no error
constexpr u32 LV_BITMAP_ONLY = 2;  // ... the bitmap of level == count_level goes to out2, nothing is counted or refused (repetition levels)
// one block on its own (widths 0..8, one byte per value); see pqd_levels.hip
constexpr size_t PQD_NOALLOC_MAX_BLOCK = 16u << 20;  // bytes of runs the entry point without allocations accepts
hipError_t lv_scratch_init(int device);
void* lv_scratch_ptr(int device);
size_t lv_scratch_bytes(size_t runs_len);
hipError_t launch_rle_decode_u8(hipStream_t st, void* scratch, int width, const u8* runs, u32 len, u8* out, u32 max_values,
                                long long* result_device);
// the same over the device's static scratch; calls from different streams are queued behind each other
hipError_t launch_rle_decode_u8_static(int device, hipStream_t st, int width, const u8* runs, u32 len, u8* out, u32 max_values,
                                       long long* result_device);

hipError_t launch_page_scan(hipStream_t st, const ChunkRec* chunks, u32 n_chunks, const PageRec* pages, u32 n_pages,
                            PageDyn* page_dyn, ChunkDyn* chunk_dyn, int* status);
// value index: per 64 levels of a chunk, the number of non-null levels in front of them (one workgroup per page)
hipError_t launch_value_index(hipStream_t st, const ChunkRec* chunks, const PageRec* pages, const PageDyn* page_dyn,
                              const u32* page_ids, u32 n_pages, const u64* page_lvl_off);
hipError_t launch_bools(hipStream_t st, const u8* src, const PageRec* pages, const u32* page_ids, const u32* btile_prefix,
                        const u32* btile_owner, u32 n_bpages, u32 n_btiles, const ChunkRec* chunks, const PageDyn* page_dyn,
                        const int* status);
// null-mask scatter of one chunk: dense values (width 1, 4 or 8 bytes) or string offsets (offsets = true, width 4,
// n_slots = levels + 1) to their level slots; dst 16-byte aligned; n_words = 64-bit words of the validity bitmap;
// n_src = entries the source array holds (indices at or beyond it are not read)
hipError_t launch_scatter(hipStream_t st, int width, bool offsets, const u8* validity, const u8* value_index, const u8* values,
                          u8* dst, u64 n_slots, u64 n_words, u32 n_src);
// strings: per-page value and char offsets within the chunk from the first pass's string counts (one wave per
// string chunk; region_prefix has n_str_chunks + 1 entries); str_pd is indexed by string region
hipError_t launch_str_page_scan(hipStream_t st, const Region* regions, const u32* region_prefix, u32 n_str_chunks,
                                const RegionTotal* totals, PageDyn* str_pd, const u32* gate);
hipError_t launch_str_check(hipStream_t st, const PageRec* pages, const ChunkRec* chunks, const PageDyn* page_dyn,
                            const Region* regions, u32 n_regions, const RegionTotal* totals, int* status);
hipError_t launch_rows(hipStream_t st, const ChunkRec* chunks, const u32* rtile_prefix, const u32* rtile_owner, const u32* chunk_ids,
                       u32 n_rchunks, u32 n_rtiles, RowTile* tiles, ChunkDyn* chunk_dyn);

// snappy: decode tiles of SN_T output bytes from the element records (otile_region / otile_first_run have one
// spare entry each); every Region.out must be 16-byte aligned; tile_done: n_otiles + 1 words, all zero (one flag
// per tile and the ticket counter)
hipError_t launch_snappy_tiles(hipStream_t st, const u8* src, const Region* regions, const u32* otile_region, u32 n_otiles,
                               const RunRec* run_tbl, const u32* otile_first_run, const RegionTotal* totals, int* status,
                               u32* tile_done);
// sizes[2*i], sizes[2*i+1] = rep / def level block bytes of page i (PageRec.val_src_off / val_len = its whole payload)
hipError_t launch_level_block_sizes(hipStream_t st, const u8* src, const PageRec* pages, const ChunkRec* chunks, u32 n_pages,
                                    int* sizes);

// ---- level encoder (pqd_encode.hip) ----------------------------------------------------------
constexpr u32 EN_T = 2048;  // level positions per encoder tile
constexpr u32 EN_E = 9;     // chain entries per tile: its first 8 positions + the end of a run that reaches into it
constexpr u32 EN_LAYOUT_CHUNK = 2048;  // elements per workgroup of the layout and write kernels
constexpr u32 EN_SEG_RECS = (EN_T / 256) * EN_E;  // 8 segments of 256 positions, EN_E walkers each
struct EncBlock {
    u64 src_off;    // first level of the block (4-byte aligned)
    u32 n;          // levels
    u32 width;      // bit width, 0..8
    u32 elem_base;  // first slot of the block in the element arrays (n / 8 + 2 slots)
    u32 pad;
    u8* out;        // int32 length + runs go here
};
struct EncArgs {
    const EncBlock* blocks;
    const u32* tile_prefix;  // n_blocks + 1
    const u32* elem_prefix;  // n_blocks + 1 (== elem_base of every block, then the total)
    const u32* lchunk_prefix;  // n_blocks + 1: layout chunks (EN_LAYOUT_CHUNK element slots each) in front of every block
    u32 n_lchunks;
    void* chunks;            // n_lchunks * 16 bytes
    u32 n_blocks, n_tiles, elem_slots;
    u32* lead;               // n_tiles
    u32* ext;                // n_tiles + 1
    void* exits;             // n_tiles * EN_E * 8 bytes
    void* entries;           // n_tiles * 8 bytes, reset to 0xFF
    u32* segs;               // n_tiles * EN_SEG_RECS: per tile, the exits of the segment walkers
    u32 *n_elems, *out_len;  // n_blocks
    u32* elems;              // elem_slots
};
hipError_t launch_rle_encode(hipStream_t st, const u8* src, const EncArgs& a);
hipError_t launch_rle_encode_layout(hipStream_t st, const u8* src, const EncArgs& a);
hipError_t launch_rle_encode_write(hipStream_t st, const u8* src, const EncArgs& a);
constexpr u32 ENC_VALUE_CHUNK = 65536;  // value bytes (strings: 16 bytes counted per string) per workgroup of k_encp_values
// one page to encode: [rep levels][def levels][PLAIN values]
struct EncPage {
    u64 values_off;   // dense LE values / one byte per bool / the chars of the strings
    u64 offsets_off;  // strings: int32[n_values + 1], 4-byte aligned
    u64 n_values;
    u32 width;        // 4 / 8 for fixed-width types, else 0
    int phys_type;
    u32 stats_kind;   // 0 signed, 1 unsigned, 2 floating point
    u32 rep_block, def_block;  // indices into the level block table, 0xFFFFFFFF = none
    // filled on the device
    u32 rep_bytes, def_bytes, pad;
    u64 val_bytes, payload_off;
    u64 min_bits, max_bits;
};
hipError_t launch_encode_pages(hipStream_t st, const u8* src, EncPage* pages, const u32* chunk_prefix, u32 n_pages, u32 n_chunks,
                               EncBlock* blocks, const u32* out_len, u8* out_base, u64* stats_partial);

}  // namespace pqd
