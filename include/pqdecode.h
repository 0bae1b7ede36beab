/*
 * pqdecode.h — C ABI of libpqdecode.so: MI355X (gfx950) decoder for uncompressed Parquet
 * DATA_PAGE (v1) column pages: hybrid RLE/bit-packed levels + PLAIN values.
 *
 * This is the boundary a cgo shim binds.  The reference (parsyl/parquet, pure Go) has no FFI;
 * each entry point below names the reference code it stands in for (paths relative to the
 * parsyl/parquet module root).  Plain pointers and sizes only; every function returns 0 or a
 * negative pqd_status and never aborts the process.
 *
 * Threading: one pqd_ctx per reader, not used concurrently (the reference reader is single
 * goroutine).  A ctx may be called from any OS thread; it selects its device on entry.
 */
#ifndef PQDECODE_H
#define PQDECODE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    PQD_OK = 0,
    PQD_E_ARG = -1,         /* bad argument */
    PQD_E_HIP = -2,         /* HIP runtime error (message in pqd_last_error) */
    PQD_E_NOMEM = -3,
    PQD_E_FORMAT = -4,      /* not a Parquet file / thrift decode failed / missing required field */
    PQD_E_UNSUPPORTED = -5, /* compressed chunk, dictionary page, INT96/FIXED, level width > 8 */
    PQD_E_CORRUPT = -6,     /* data the reference would fail or panic on (short read, bad run, negative length) */
    PQD_E_NONCANONICAL = -7 /* bool/string page whose value bytes are not exactly consumed; the
                               reference would carry the remainder into the next page */
} pqd_status;

/* Parquet physical types (schema/parquet.thrift Type). */
enum {
    PQD_BOOLEAN = 0, PQD_INT32 = 1, PQD_INT64 = 2, PQD_INT96 = 3, PQD_FLOAT = 4, PQD_DOUBLE = 5,
    PQD_BYTE_ARRAY = 6, PQD_FIXED_LEN_BYTE_ARRAY = 7
};

/* One column chunk (the reference's parquet.Page, parquet.go:28-34, which despite its name
 * describes a chunk) plus what the generated Field carries: physical type and max levels
 * (fields.go:38-57, 147-152). */
typedef struct {
    int32_t phys_type;
    int32_t max_def;     /* 0 → RequiredField.DoRead, >0 → OptionalField.DoRead */
    int32_t max_rep;
    int32_t first_page;  /* index into the page array */
    int32_t n_pages;
    int32_t column;      /* leaf index in the file schema (informational; set by pqd_file_descriptors) */
    int64_t num_values;  /* parquet.Page.N: levels in the chunk, nulls included (parquet.go:336) */
    int32_t codec;       /* parquet.Page.Codec (parquet.go:32): 0 UNCOMPRESSED, 1 SNAPPY; others are not decoded */
    int32_t reserved;
} pqd_chunk_desc;

/* One data page as the reference's page loops see it (fields.go:114-135, 275-317). */
typedef struct {
    int64_t src_off;     /* offset of the page payload (first byte after the thrift PageHeader) */
    int32_t payload_len; /* bytes pageData reads: uncompressed_page_size, or compressed_page_size under a codec */
    int32_t num_values;  /* DataPageHeader.num_values */
    int32_t rep_bytes;   /* size of the repetition-level block incl. its 4-byte length, 0 if none; -1 = read it from src */
    int32_t def_bytes;   /* same for definition levels */
} pqd_page_desc;

/* Which decoded array of a chunk. */
typedef enum {
    PQD_OUT_DEFS = 0,         /* uint8[n_levels]  — OptionalField.Defs (fields.go:157, :308) */
    PQD_OUT_REPS = 1,         /* uint8[n_levels]  — OptionalField.Reps (fields.go:158, :300) */
    PQD_OUT_VALUES = 2,       /* dense non-null values: LE fixed width, or 1 byte per bool — the f.vals of
                                 template_required.go:31-33 / template_optional.go:48-50 / parquet.go:455-484 */
    PQD_OUT_OFFSETS = 3,      /* int32[n_values+1] string offsets into CHARS (template_string.go:47-58) */
    PQD_OUT_CHARS = 4,        /* concatenated string bytes */
    PQD_OUT_VALIDITY = 5,     /* bitmap, bit i (LSB first) = (defs[i] == max_def)   (fields.go:220-228 per level) */
    PQD_OUT_PAGE_NONNULL = 6, /* int64[n_pages] — the `sizes` of DoRead (fields.go:124, :311-312) */
    PQD_OUT_ROW_LEVEL_OFF = 7,/* int32[n_rows+1] — first level of each row (write_repeated.go:62-80); only for
                                 repeated columns (max_rep > 0): otherwise row r is level r and its value index is
                                 the number of set validity bits before r */
    PQD_OUT_ROW_VALUE_OFF = 8,/* int32[n_rows+1] — first value of each row; repeated columns only */
    PQD_OUT_VALUE_INDEX = 9   /* uint32[ceil(n_levels/64)+1] — per 64 levels, the non-null levels in front of them (valsFromDefs
                                 per level, fields.go:216-228): the value of non-null level i of an optional column, which the
                                 generated write functions consume row by row (performance/parquet.go:104-113), is
                                 values[VALUE_INDEX[i/64] + popcount(VALIDITY bits of word i/64 below bit i%64)]; the last
                                 entry is n_values.  Columns with definition levels only */
} pqd_output;

typedef struct {
    int64_t n_levels;   /* entries in DEFS (and REPS when max_rep > 0) */
    int64_t n_values;   /* non-null values (required chunk: num_values) */
    int64_t value_bytes;/* bytes in VALUES (fixed width: n_values*w; bool: n_values; string: 0) */
    int64_t chars_bytes;/* bytes in CHARS */
    int64_t n_rows;     /* rows spanned by the chunk (entries-1 of the row offset arrays) */
    int64_t n_pages;
} pqd_chunk_info;

typedef struct pqd_ctx pqd_ctx;
typedef struct pqd_plan pqd_plan;

/* ---- context ------------------------------------------------------------------------------ */
int pqd_create(int device, pqd_ctx** out);
void pqd_destroy(pqd_ctx* ctx);
const char* pqd_last_error(pqd_ctx* ctx);

/* Pinned host staging owned by the library, so a Go caller never hands the library a Go pointer
 * that outlives the call. */
void* pqd_host_alloc(pqd_ctx* ctx, size_t bytes);
void pqd_host_free(pqd_ctx* ctx, void* p);
/* Device buffer for file bytes (device-resident path). */
void* pqd_device_alloc(pqd_ctx* ctx, size_t bytes);
void pqd_device_free(pqd_ctx* ctx, void* p);
int pqd_copy_to_device(pqd_ctx* ctx, void* dst_device, const void* src_host, size_t bytes);
int pqd_copy_to_host(pqd_ctx* ctx, void* dst_host, const void* src_device, size_t bytes);

/* ---- decode: replaces the bodies of the six generated Field.Read methods -------------------
 * (cmd/parquetgen/gen/template_required.go:25-35, template_optional.go:42-52, template_string.go:41-60,
 *  template_string_optional.go:62-81, template_bool.go:38-46, template_bool_optional.go:24-33)
 * together with RequiredField/OptionalField.DoRead (fields.go:114-135, 275-317), readLevels
 * (fields.go:448-457), rle.Read (internal/rle/rle.go:159-290), bitpack.Unpack
 * (internal/bitpack/bitpack.go:76-141) and GetBools (parquet.go:455-505), for a whole batch of
 * column chunks — normally every chunk of one or more row groups, i.e. one readRowGroup
 * (template.go:320-351) per call.
 *
 * pqd_plan_create does the host-side part: validates the descriptors, reads the level-block
 * lengths (rep_bytes/def_bytes == -1) from `src_host`, sizes the outputs and uploads the tables.
 * `src_host` may be NULL when every page carries rep_bytes/def_bytes.  `src_len` is the length of
 * the byte range the page offsets index into.
 */
int pqd_plan_create(pqd_ctx* ctx, const pqd_chunk_desc* chunks, size_t n_chunks,
                    const pqd_page_desc* pages, size_t n_pages,
                    const void* src_host, size_t src_len, pqd_plan** out);
void pqd_plan_destroy(pqd_plan* plan);

/* Runs the decode kernels over `src_device` (device pointer to the same `src_len` bytes the plan was
 * built for).  Asynchronous on the ctx stream; outputs stay in HBM.  May be called repeatedly. */
int pqd_plan_run(pqd_plan* plan, const void* src_device);
/* Waits for the last run and returns its status (PQD_OK or the first decode error). */
int pqd_plan_sync(pqd_plan* plan);
/* Device time of the last completed run in milliseconds (first kernel start → last kernel end),
 * and of its dominant kernel (the PLAIN value gather), both from HIP events on the ctx stream. */
int pqd_plan_last_ms(pqd_plan* plan, float* total_ms, float* plain_copy_ms);
/* The same for an earlier run: runs_back = 0 is the last run, 1 the one before, ... (the plan keeps
 * the events of its last 256 runs, so a caller can queue many runs and read all their timings after
 * one sync). */
int pqd_plan_run_ms(pqd_plan* plan, int runs_back, float* total_ms, float* plain_copy_ms);
/* Algorithmic bytes of one run: payload bytes read + decoded bytes the reference materialises
 * (defs, reps, values / bools / string offsets+chars); and the part moved by the value gather. */
int pqd_plan_bytes(pqd_plan* plan, int64_t* bytes_in, int64_t* bytes_out, int64_t* plain_copy_bytes);

/* Results of the last synced run. */
int pqd_plan_chunk_info(pqd_plan* plan, size_t chunk, pqd_chunk_info* info);
const void* pqd_plan_device_ptr(pqd_plan* plan, size_t chunk, pqd_output which);
/* Copies an output array to host memory; `cap` is the size of `dst` in bytes. Returns bytes copied
 * (>= 0) or a negative status. */
int64_t pqd_plan_fetch(pqd_plan* plan, size_t chunk, pqd_output which, void* dst, size_t cap);

/* Null-mask scatter of one chunk of the last run: its values at their level slots, for consumers that want one
 * entry per level (row, for a column that is not repeated) instead of the dense f.vals plus levels that the
 * reference's Scan consumes (valsFromDefs fields.go:216-228; performance/parquet.go:104-113).
 * Fixed width and bool: dst[i] = the value of level i, 0 where validity bit i is clear; n_levels entries of
 * 4 / 8 / 1 bytes.  BYTE_ARRAY: int32[n_levels + 1] offsets into CHARS, dst[i + 1] - dst[i] = length of the
 * string of level i and 0 for a null.  Columns with definition levels only (PQD_E_ARG otherwise).
 * `dst_device` is device memory of the caller, 16-byte aligned, at least n_levels * w (strings:
 * (n_levels + 1) * 4) bytes.  Queued on the decoder's stream behind the last run; pqd_plan_sync waits for it. */
int pqd_plan_scatter(pqd_plan* plan, size_t chunk, void* dst_device, size_t dst_bytes);

/* Diagnostics: BYTE_ARRAY column chunks of the last synced run that the single pass over the string bytes gave up and
 * the exact two-pass kernels decoded instead (contents whose bytes look like length prefixes, corrupt pages).  The
 * results are the same either way; a large number explains a slow run.  0 when PQD_STR_PATH=twopass sends every chunk
 * through the two passes. */
int64_t pqd_plan_string_fallbacks(pqd_plan* plan);

/* Diagnostics: when the plan was created with PQD_LV_STAMPS set in the environment, the level kernel of the
 * last run left 16 wall-clock stamps (100 MHz) per workgroup; copies them to `dst` and returns the bytes
 * (0 when the plan has none).  With PQD_STR_STAMPS set instead, the same for the single-pass string kernel. */
int64_t pqd_plan_debug_stamps(pqd_plan* plan, void* dst, size_t cap);

/* Convenience: host buffer in, results on device; = copy_to_device + plan_create + run + sync.
 * Only the byte range that the described pages cover (first payload byte to last, the page headers between them
 * included) is copied to the device, so decoding a file row group by row group, as readRowGroup does
 * (template.go:320-351), moves every byte of it once.  The plan owns the device copy. */
int pqd_decode(pqd_ctx* ctx, const pqd_chunk_desc* chunks, size_t n_chunks,
               const pqd_page_desc* pages, size_t n_pages,
               const void* src_host, size_t src_len, pqd_plan** out);
/* Bytes pqd_decode copied from host to device for this plan (0 for a plan made by pqd_plan_create). */
int64_t pqd_plan_h2d_bytes(pqd_plan* plan);
/* Column chunks of the plan whose repetition level blocks are byte for byte those of an earlier chunk of the batch
 * (leaves under the same repeated groups, e.g. Friends.ID and Friends.Age of README.md:146-159): their PQD_OUT_REPS and
 * PQD_OUT_ROW_LEVEL_OFF are that chunk's arrays (the same device pointers), decoded and written once.  Found only for
 * blocks the planner located through `src_host` itself; PQD_SHARE_REPS=0 in the environment turns it off. */
int64_t pqd_plan_shared_reps(pqd_plan* plan);

/* ---- standalone level decode: rle.Read (internal/rle/rle.go:159-195) for one block ----------
 * `block` = int32 LE length + runs, in host memory.  width 0..32; out_elem_size 1 (width <= 8),
 * 2 (<= 16) or 4.  Decodes at most `max_values` values into `out_host`; *n_decoded receives the
 * number the block holds (bit-packed tails count in full, as in the reference). */
int pqd_rle_decode(pqd_ctx* ctx, int width, const void* block, size_t block_len,
                   int out_elem_size, void* out_host, size_t max_values, int64_t* n_decoded);

/* ---- the same decode for callers that cannot allocate -----------------------------------------
 * No counterpart in the reference.  It is for callers that must not call hipMalloc: code under HIP graph
 * capture, tight serving loops, harnesses that forbid allocation.  This entry point allocates nothing (no
 * hipMalloc / hipFree, no host allocation) and does not synchronise: it queues three kernels on the context's
 * stream and returns.  Inputs and outputs are the caller's DEVICE buffers:
 *   runs_device    the hybrid runs of one level block WITHOUT the 4-byte length in front; 8 readable bytes of
 *                  slack behind runs_len are not required
 *   out_device     uint8[max_values]; width 0..8
 *   result_device  int64[2]: {values the block holds (saturated at 2^32 - 1; min(that, max_values) were
 *                  written), 0 or a negative pqd_status}
 * The tables of the decode live in a static device array of the library (about 50 KB per GPU, held until the
 * process ends), which limits a block to PQD_NOALLOC_MAX_BLOCK_BYTES of runs; larger blocks are refused with
 * PQD_E_ARG.  Every call on one GPU shares that array.  Calls on one context are ordered by its stream; a call
 * from another context of the same GPU is queued behind the call before it (an event the library keeps per GPU),
 * so calls from several contexts or host threads do not overlap on the array, they take turns.  Code that
 * launches the same work on streams of its own (a replayed graph) must keep that order itself: replays that
 * overlap on one GPU race.  Read the results with pqd_copy_to_host, which waits for the stream. */
#define PQD_NOALLOC_MAX_BLOCK_BYTES (16u << 20)
int pqd_rle_decode_noalloc(pqd_ctx* ctx, int width, const void* runs_device, size_t runs_len,
                           void* out_device, size_t max_values, int64_t* result_device);

/* ---- standalone level ENCODE: rle.(*RLE).Write + Bytes (internal/rle/rle.go:48-156) for one block ----
 * The mirror of pqd_rle_decode and the first piece of the write path: `levels` (one byte each, host memory)
 * become int32 LE length + hybrid runs, byte for byte what the reference encoder writes: at a group
 * boundary eight equal values open an RLE run that absorbs every following equal value, other values are
 * bit-packed in groups of eight (the last one zero padded) under headers of at most 63 groups.
 * width 0..8 (the reference packs widths 1..4 only; 5..8 follow the format text).  *out_len receives the
 * encoded size even when `out_cap` is too small (PQD_E_ARG then). */
int pqd_rle_encode(pqd_ctx* ctx, int width, const void* levels_host, size_t n, void* out_host, size_t out_cap,
                   int64_t* out_len);

/* ---- page payloads of the write path (RequiredField.DoWrite fields.go:95-111, OptionalField.DoWrite
 * fields.go:232-271, codec UNCOMPRESSED): per page [repetition levels][definition levels][PLAIN values],
 * the levels through the encoder above, the values as the generated Write methods serialise them
 * (binary.Write little endian: template_required.go:37-49, template_optional.go:96-114; bools 8 per byte
 * LSB first: template_bool.go; strings int32 length + bytes: template_string.go:62-78).  The host frames
 * the payloads with thrift page headers, as the reference does. */
typedef struct {
    int32_t phys_type;
    int32_t is_unsigned;  /* converted type UINT_32 / UINT_64: statistics compare unsigned */
    int32_t max_def, max_rep;
    int64_t n_levels;     /* len(Defs); for a required column the number of values */
    int64_t n_values;     /* non-null values */
    int64_t defs_off, reps_off; /* uint8 per level, offsets into `src`, 4-byte aligned; ignored when the max level is 0 */
    int64_t values_off;   /* dense little-endian values (aligned to their width), one byte per bool, or the string bytes */
    int64_t offsets_off;  /* strings: int32[n_values + 1] into the string bytes, 4-byte aligned */
} pqd_enc_page;
typedef struct {
    int64_t payload_off, payload_len; /* where the page payload lies in `out` */
    int32_t rep_bytes, def_bytes;     /* the repLen / defLen DoWrite hands to WritePageHeader */
    int64_t null_count;               /* levels below max_def (the nils of the generated stats types) */
    uint8_t min[8], max[8];           /* fixed-width types: as the stats types keep them (min from the type's
                                         maximum, max from ZERO: template_required.go:77-96) */
    int32_t minmax_len;               /* 4 / 8, or 0 = Min()/Max() return nil (no non-null value; bools, strings) */
    int32_t reserved;
} pqd_enc_result;
int pqd_encode_pages(pqd_ctx* ctx, const pqd_enc_page* pages, size_t n, const void* src_host, size_t src_len,
                     void* out_host, size_t out_cap, pqd_enc_result* results, int64_t* out_len);

/* ---- SNAPPY pages: pageData's SNAPPY arm (fields.go:361-371, snappy.Decode of golang/snappy) ------
 * pqd_decode handles SNAPPY chunks by itself; these are its steps for callers that keep the file on
 * the device.  A block is one page payload in snappy's raw block format (uvarint decoded length, then
 * literal / copy elements). */
typedef struct {
    int64_t src_off;     /* in: first byte of the compressed page payload */
    int64_t comp_len;    /* in: PageHeader.compressed_page_size */
    int64_t dst_off;     /* in (decode): where the decoded bytes go; base + dst_off must be 16-byte aligned */
    int64_t decoded_len; /* out of pqd_snappy_sizes: the length the preamble declares (decodedLen) */
    int32_t hdr_len;     /* out of pqd_snappy_sizes: bytes of the preamble */
    int32_t reserved;
} pqd_snappy_block;
/* Host only: reads the uvarint preamble of every block.  PQD_E_CORRUPT as snappy.ErrCorrupt. */
int pqd_snappy_sizes(const void* src_host, size_t src_len, pqd_snappy_block* blocks, size_t n);
/* Decodes n blocks from device memory to device memory and waits for them.  PQD_E_CORRUPT where
 * snappy.Decode returns ErrCorrupt.  Copies may reach back as far as the format allows (any offset up to
 * the bytes written so far, all three copy tags), as in snappy.Decode. */
int pqd_snappy_decode(pqd_ctx* ctx, const pqd_snappy_block* blocks, size_t n, const void* src_device, size_t src_len,
                      void* dst_device, size_t dst_len);
/* Fills rep_bytes / def_bytes of pages whose payload exists only on the device (decoded pages): the
 * int32 length in front of each level block (rle.go:161-164, fields.go:295-309). */
int pqd_level_block_sizes(pqd_ctx* ctx, const pqd_chunk_desc* chunks, size_t n_chunks, pqd_page_desc* pages, size_t n_pages,
                          const void* src_device, size_t src_len);

/* ---- file walk: the host-side prerequisite (parquet.go:349-378 ReadMetaData/PageHeader,
 * parquet.go:322-346 Pages, template.go:320-351 readRowGroup's chunk order) ------------------- */
typedef struct pqd_file pqd_file;

typedef struct {
    int32_t phys_type;
    int32_t converted_type; /* -1 if absent */
    int32_t max_def;
    int32_t max_rep;
    int32_t leaf_repetition; /* 0 required, 1 optional, 2 repeated */
    int32_t reserved;
} pqd_column_info;

/* Parses the footer and walks every page header of every column chunk sequentially from byte 4,
 * exactly as the reference reader consumes the file (it never seeks: template.go:263). */
int pqd_file_open(const void* buf, size_t len, pqd_file** out, char* err, size_t err_cap);
void pqd_file_close(pqd_file* f);
int64_t pqd_file_num_rows(const pqd_file* f);
int32_t pqd_file_num_columns(const pqd_file* f);
int32_t pqd_file_num_row_groups(const pqd_file* f);
int64_t pqd_file_row_group_rows(const pqd_file* f, int32_t rg);
int pqd_file_column(const pqd_file* f, int32_t col, pqd_column_info* info);
const char* pqd_file_column_path(const pqd_file* f, int32_t col); /* dotted, as Field.Name() */
/* Chunk and page descriptors of row groups [rg_first, rg_first+rg_count); chunk c of row group g
 * is entry (g-rg_first)*num_columns + c.  The returned arrays belong to the file object and stay
 * valid until the next call or pqd_file_close. */
int pqd_file_descriptors(pqd_file* f, int32_t rg_first, int32_t rg_count,
                         const pqd_chunk_desc** chunks, size_t* n_chunks,
                         const pqd_page_desc** pages, size_t* n_pages);
/* Total page headers walked (the reference's PageHeaders, parquet.go:382-394). */
int64_t pqd_file_num_pages(const pqd_file* f);
/* One walked page header, the equivalent of `parquetgen -pageheaders` (cmd/parquetgen/main.go:51-80):
 * out = {type, uncompressed_page_size, compressed_page_size, num_values, encoding,
 *        definition_level_encoding, repetition_level_encoding, header length in bytes}. */
int pqd_file_page_header(const pqd_file* f, int64_t page, int32_t out[8], int64_t* header_pos);
/* CompressionCodec of a column chunk (0 UNCOMPRESSED, 1 SNAPPY; others are walked but not decoded). */
int pqd_file_chunk_codec(const pqd_file* f, int32_t rg, int32_t chunk);
/* What Metadata.Pages (parquet.go:322-346) hands the Field for one chunk, plus where the walk found it:
 * out = {leaf column index, ColumnMetaData.num_values (Page.N), total_compressed_size (Page.Size),
 *        file position of the chunk's first page header, index of its first walked page, pages walked}. */
int pqd_file_chunk_meta(const pqd_file* f, int32_t rg, int32_t chunk, int64_t out[6]);

/* ---- multi-GPU: whole row groups per GPU ---------------------------------------------------------
 * Nothing in the reference corresponds to this (it is one goroutine on one host); what makes it possible is
 * that readRowGroup starts every row group from fresh Field objects (cmd/parquetgen/gen/template.go:329), so row
 * groups are independent.  GPUs take contiguous ranges of row groups balanced by payload bytes, decode them with
 * no data-path exchange, and only exchange their row counts to learn their global row offsets. */

/* Contiguous ranges [first[r], first[r] + count[r]) of n weighted items for `world` ranks: cut r lies at the item
 * boundary nearest to r/world of the total weight.  Host only. */
int pqd_shard_assign(const int64_t* weights, int32_t n, int32_t world, int32_t* first, int32_t* count);
/* The same over the row groups of a file, weighted by their chunks' total_compressed_size (parquet.go:197). */
int pqd_file_shard(const pqd_file* f, int32_t world, int32_t* first, int32_t* count);

/* One process, several GPUs: one context and one host thread per device (a device may be listed more than once:
 * then its shards run on separate contexts). */
typedef struct pqd_multi pqd_multi;
int pqd_multi_create(const int* devices, int n, pqd_multi** out);
void pqd_multi_destroy(pqd_multi* m);
/* Walks the file, assigns its row groups, decodes every shard on its device (pqd_decode, all shards at once) and
 * scans the shards' row counts.  Returns the first shard error, if any. */
int pqd_multi_decode_file(pqd_multi* m, const void* buf, size_t len);
int pqd_multi_shards(const pqd_multi* m);
/* Row group range, rows, global row offset (rows of the shards before it) and decoded plan of one shard; the plan
 * (NULL for a shard without row groups) belongs to the object and lives until the next decode or destroy. */
int pqd_multi_shard(const pqd_multi* m, int shard, int32_t* rg_first, int32_t* rg_count, int64_t* rows, int64_t* row_offset,
                    pqd_plan** plan);
const char* pqd_multi_last_error(const pqd_multi* m);

/* One process per GPU: the row counts go through RCCL (ncclAllGather of one int64 per rank, xGMI inside a node).
 * librccl is loaded on first use (PQD_E_UNSUPPORTED when it is absent).  Rank 0 makes the 128-byte id and hands it
 * to the other ranks by whatever means the host program has.  A communicator owns its RCCL communicator, a stream and
 * 8 (world + 1) bytes of device and of pinned host memory, so that a gather allocates and creates nothing. */
int pqd_rccl_unique_id(void* id128);
int pqd_rccl_comm_create(int device, int rank, int world, const void* id128, void** comm);
void pqd_rccl_comm_destroy(void* comm);
/* all_rows: host array of `world` entries; *row_offset = rows of the ranks in front of this one.  Waits for the result. */
int pqd_rccl_gather_rows(void* comm, int64_t local_rows, int64_t* all_rows, int64_t* row_offset);

#ifdef __cplusplus
}
#endif
#endif /* PQDECODE_H */
