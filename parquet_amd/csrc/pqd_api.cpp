// C ABI of libpqdecode.so (see include/pqdecode.h): context, decode plans, result access.
// The host side here only plans (descriptor validation, table building, output sizing) and
// launches; every byte of decode work happens in the kernels of pqd_kernels.hip.  There is no CPU
// decode path: without a working HIP device every entry point that needs one fails with PQD_E_HIP.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/pqdecode.h"
#include "pqd_device.h"

using namespace pqd;

struct pqd_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;  // side stream of the first pass over the strings
    hipStream_t stream3 = nullptr;  // side stream of the PLAIN value gather
    bool overlap = true;            // PQD_OVERLAP=0 keeps everything on one stream
    bool str_after_levels = false;  // PQD_STR_AFTER_LEVELS=1: diagnostics, see enqueue_run
    bool str_fused = true;          // PQD_STR_PATH=twopass: every string chunk through the two passes (the single pass's fallback)
    bool str_giveup = false;        // PQD_STR_PATH=giveup: the single pass runs but finds every chunk given up already (tests of the hand-over)
    bool graph = false;             // PQD_GRAPH=1 replays chain batches as a captured HIP graph
    bool levels_first = false;      // PQD_LEVELS_FIRST=1: diagnostics, see enqueue_run
    bool rows_first = true;         // PQD_ROWS_FIRST=0: the row offsets behind the value index, as they once were
    std::string err;
    // Device blocks of destroyed plans, kept for the next plan of this context: a reader makes one plan per row group, all
    // of about the same size, and hipMalloc / hipFree cost a millisecond a pair and synchronise the device.
    struct Block { void* p; size_t bytes; };
    std::vector<Block> pool;                 // free blocks (at most POOL_MAX)
    std::vector<Block> live;                 // blocks handed out by pool_alloc
    static constexpr size_t POOL_MAX = 4;
    u32 copy_grid = 0x7fffffffu;  // one block per 4 KiB tile; PQD_COPY_GRID caps it (then blocks grid-stride)
};

namespace {

// a block of at least `bytes` from the context's pool (the smallest that fits and is not more than twice as large), else
// from hipMalloc; null on failure
void* pool_alloc(pqd_ctx* ctx, size_t bytes) {
    if (bytes == 0) bytes = 256;
    size_t best = ctx->pool.size();
    for (size_t i = 0; i < ctx->pool.size(); i++)
        if (ctx->pool[i].bytes >= bytes && ctx->pool[i].bytes / 2 <= bytes && (best == ctx->pool.size() || ctx->pool[i].bytes < ctx->pool[best].bytes))
            best = i;
    pqd_ctx::Block b{nullptr, bytes};
    if (best < ctx->pool.size()) {
        b = ctx->pool[best];
        ctx->pool.erase(ctx->pool.begin() + (long)best);
    } else if (hipMalloc(&b.p, bytes) != hipSuccess) {
        // memory may be held by the pool: give it back and try once more
        for (auto& f : ctx->pool) (void)hipFree(f.p);
        ctx->pool.clear();
        if (hipMalloc(&b.p, bytes) != hipSuccess) return nullptr;
    }
    ctx->live.push_back(b);
    return b.p;
}
// gives a block of pool_alloc back; the caller has waited for all work that uses it
void pool_release(pqd_ctx* ctx, void* p) {
    if (!p) return;
    for (size_t i = 0; i < ctx->live.size(); i++)
        if (ctx->live[i].p == p) {
            const pqd_ctx::Block b = ctx->live[i];
            ctx->live.erase(ctx->live.begin() + (long)i);
            if (ctx->pool.size() >= pqd_ctx::POOL_MAX) {  // the smallest goes
                size_t m = 0;
                for (size_t k = 1; k < ctx->pool.size(); k++) if (ctx->pool[k].bytes < ctx->pool[m].bytes) m = k;
                if (ctx->pool[m].bytes >= b.bytes) { (void)hipFree(b.p); return; }
                (void)hipFree(ctx->pool[m].p);
                ctx->pool.erase(ctx->pool.begin() + (long)m);
            }
            ctx->pool.push_back(b);
            return;
        }
    (void)hipFree(p);  // not from the pool
}

int bits_len(unsigned v) { int n = 0; while (v) { n++; v >>= 1; } return n; }  // math/bits.Len
size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

int fail(pqd_ctx* ctx, int code, const char* what, hipError_t e = hipSuccess) {
    if (ctx) {
        ctx->err = what;
        if (e != hipSuccess) { ctx->err += ": "; ctx->err += hipGetErrorString(e); }
    }
    return code;
}

#define HIP_TRY(ctx, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return fail((ctx), PQD_E_HIP, #call, e_); } while (0)

constexpr int N_OUT = 10;  // pqd_output values
struct OutPtrs { u8* p[N_OUT]; size_t cap[N_OUT]; };

}  // namespace

struct pqd_plan {
    pqd_ctx* ctx = nullptr;
    size_t src_len = 0;
    u8* owned_src = nullptr;       // device copy made by pqd_decode
    int64_t h2d_bytes = 0;         // bytes pqd_decode copied to the device for this plan
    int n_shared_rep = 0;          // chunks whose repetition levels are another chunk's (byte-identical blocks)
    const u8* last_src = nullptr;

    // host copies of the tables
    std::vector<ChunkRec> chunks;
    std::vector<PageRec> pages;
    std::vector<Region> lregions, sregions;
    std::vector<LvTile> lv_order;  // launch order of the level super-tiles
    std::vector<u32> stile_chain_prefix, sgroup_prefix, ssuper_prefix;
    std::vector<Seg> segs;
    std::vector<u32> seg_blk_prefix, seg_grp;
    std::vector<u32> bool_pages, btile_prefix, row_chunks, rtile_prefix;
    std::vector<u32> btile_owner, rtile_owner;  // per tile: its index into bool_pages / row_chunks (no search on the device)
    std::vector<u32> stile_region;  // region of every string chain tile
    std::vector<u32> sunit_region;   // single string pass: region of every unit
    std::vector<StrUnit> sunits;     // ... and the units in launch order
    std::vector<u32> vidx_pages;    // pages of the chunks that have definition levels
    std::vector<u64> page_lvl_off;  // level index of every page's first level within its chunk
    u32* d_vidx_pages = nullptr; u64* d_lvl_off = nullptr;
    std::vector<OutPtrs> outs;
    std::vector<int> chunk_width;

    // device
    u8* arena = nullptr;
    size_t arena_bytes = 0;
    ChunkRec* d_chunks = nullptr; PageRec* d_pages = nullptr;
    Region *d_lregions = nullptr, *d_sregions = nullptr;
    u32* d_stile_chain_prefix = nullptr;
    u32 *d_sgroup_prefix = nullptr, *d_ssuper_prefix = nullptr;
    Seg* d_segs = nullptr; u32* d_seg_blk_prefix = nullptr; u32* d_seg_grp = nullptr;
    u32 *d_bool_pages = nullptr, *d_btile_prefix = nullptr;
    u32 *d_row_chunks = nullptr, *d_rtile_prefix = nullptr;
    u32 *d_btile_owner = nullptr, *d_rtile_owner = nullptr;
    ExitRec *d_sexit = nullptr, *d_sgexit = nullptr;
    TileEntry *d_stile_entry = nullptr, *d_sgroup_entry = nullptr;
    ChainArgs schain = {};
    LvArgs lv = {};
    RegionTotal* d_stotals = nullptr;
    std::vector<u32> str_chunk_prefix;  // first string region (page) of every BYTE_ARRAY chunk, and the total
    u32* d_str_chunk_prefix = nullptr;
    PageDyn* d_str_pd = nullptr;        // per string region: where the page's strings go in the chunk's arrays
    PageDyn* d_page_dyn = nullptr; ChunkDyn* d_chunk_dyn = nullptr; RowTile* d_row_tiles = nullptr;
    int* d_status = nullptr;
    u8 *ff_begin = nullptr, *zero_begin = nullptr;  // per-run reset ranges
    size_t ff_bytes = 0, zero_bytes = 0;

    u32 lv_tile = LV_T_SMALL_BATCH;  // bytes per super-tile of the level decode: which build of k_levels runs
    u32 n_lvtiles = 0, n_stiles_chain = 0, n_btiles = 0, n_rtiles = 0, n_copy_blocks = 0;
    int64_t bytes_in = 0, copy_bytes = 0;

    // ring of per-run event sets: {run start, run end, value gather start, value gather end}
    static constexpr int EV_RING = 256;
    hipEvent_t ev[EV_RING][4] = {};
    hipEvent_t ev_fork = nullptr;  // orders the value gather's side stream after the run start
    hipEvent_t ev_join = nullptr;  // ... the strings' marking pass after their first pass
    hipEvent_t ev_vidx = nullptr;  // ... the value index on a side stream
    hipEvent_t ev_join_copy = nullptr;  // ... and the run's end after the gather
    hipEvent_t ev_levels = nullptr;     // the page scan is done: value index on the side stream; diagnostics (PQD_STR_AFTER_LEVELS)
    hipGraphExec_t gexec = nullptr;  // captured run (see pqd_plan_run)
    const u8* gsrc = nullptr;        // the source pointer the graph was captured with
    uint64_t runs = 0;
    bool ran = false, synced = false;
    int last_status = 0;
    std::vector<ChunkDyn> h_chunk_dyn;
    int h_status[4] = {0, 0, 0, 0};
    std::vector<u32> h_str_failed;  // per BYTE_ARRAY chunk: given up by the single pass in the last synced run
};

namespace {

struct ArenaPlan {
    size_t total = 0;
    size_t take(size_t bytes) { size_t o = total; total += align_up(bytes ? bytes : 1, 256); return o; }
};

template <class T>
int upload(pqd_ctx* ctx, T* dst, const std::vector<T>& v) {
    if (v.empty()) return PQD_OK;
    HIP_TRY(ctx, hipMemcpyAsync(dst, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
    return PQD_OK;
}

int map_status(int st) {
    switch (st) {
    case 0: return PQD_OK;
    case ST_CORRUPT: return PQD_E_CORRUPT;
    case ST_NONCANONICAL: return PQD_E_NONCANONICAL;
    case ST_INTERNAL: return PQD_E_HIP;
    default: return PQD_E_UNSUPPORTED;
    }
}

}  // namespace

extern "C" {

int pqd_create(int device, pqd_ctx** out) {
    if (!out) return PQD_E_ARG;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || device < 0 || device >= n) return PQD_E_HIP;
    pqd_ctx* ctx = new pqd_ctx();
    ctx->device = device;
    // Stream priorities (PQD_PRIO=main,strings,gather; smaller = more urgent, clamped to what the device offers).  All
    // equal: with the main stream (the level decode and the chain of short kernels behind it) ahead of the string pass the
    // batch of all column kinds took 0.8 % longer, with the strings or the gather ahead of the main stream 2 % longer
    // (then the short kernels run alone at the end of the batch); the gather's priority makes no difference.
    int prio[3] = {0, 0, 0};
    if (const char* pr = getenv("PQD_PRIO")) (void)sscanf(pr, "%d,%d,%d", &prio[0], &prio[1], &prio[2]);
    int p_least = 0, p_greatest = 0;
    if (hipSetDevice(device) == hipSuccess) (void)hipDeviceGetStreamPriorityRange(&p_least, &p_greatest);
    for (int& q : prio) q = q < p_greatest ? p_greatest : q > p_least ? p_least : q;
    if (hipSetDevice(device) != hipSuccess ||
        hipStreamCreateWithPriority(&ctx->stream, hipStreamNonBlocking, prio[0]) != hipSuccess ||
        hipStreamCreateWithPriority(&ctx->stream2, hipStreamNonBlocking, prio[1]) != hipSuccess ||
        hipStreamCreateWithPriority(&ctx->stream3, hipStreamNonBlocking, prio[2]) != hipSuccess) {
        if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
        if (ctx->stream2) (void)hipStreamDestroy(ctx->stream2);
        delete ctx;
        return PQD_E_HIP;
    }
    // device address of the static scratch of pqd_rle_decode_noalloc, looked up once per device
    if (lv_scratch_init(device) != hipSuccess) {
        (void)hipStreamDestroy(ctx->stream);
        (void)hipStreamDestroy(ctx->stream2);
        (void)hipStreamDestroy(ctx->stream3);
        delete ctx;
        return PQD_E_HIP;
    }
    const char* g = getenv("PQD_COPY_GRID");
    if (g && atoi(g) > 0) ctx->copy_grid = (u32)atoi(g);
    const char* ov = getenv("PQD_OVERLAP");
    if (ov) ctx->overlap = atoi(ov) != 0;
    const char* sal = getenv("PQD_STR_AFTER_LEVELS");
    if (sal) ctx->str_after_levels = atoi(sal) != 0;
    const char* sp = getenv("PQD_STR_PATH");
    if (sp) ctx->str_fused = strcmp(sp, "twopass") != 0;
    if (sp) ctx->str_giveup = strcmp(sp, "giveup") == 0;
    const char* gr = getenv("PQD_GRAPH");
    if (gr) ctx->graph = atoi(gr) != 0;
    const char* lf = getenv("PQD_LEVELS_FIRST");
    if (lf) ctx->levels_first = atoi(lf) != 0;
    const char* rf = getenv("PQD_ROWS_FIRST");
    if (rf) ctx->rows_first = atoi(rf) != 0;
    *out = ctx;
    return PQD_OK;
}

void pqd_destroy(pqd_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    for (auto& b : ctx->pool) (void)hipFree(b.p);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    if (ctx->stream2) (void)hipStreamDestroy(ctx->stream2);
    if (ctx->stream3) (void)hipStreamDestroy(ctx->stream3);
    delete ctx;
}

const char* pqd_last_error(pqd_ctx* ctx) { return ctx ? ctx->err.c_str() : "no context"; }

void* pqd_host_alloc(pqd_ctx* ctx, size_t bytes) {
    if (!ctx) return nullptr;
    void* p = nullptr;
    if (hipSetDevice(ctx->device) != hipSuccess || hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess)
        return nullptr;
    return p;
}
void pqd_host_free(pqd_ctx* ctx, void* p) { if (ctx && p) (void)hipHostFree(p); }

void* pqd_device_alloc(pqd_ctx* ctx, size_t bytes) {
    if (!ctx) return nullptr;
    void* p = nullptr;
    // 64 bytes of slack: the kernels read whole 8/16-byte words that may straddle the last payload byte
    if (hipSetDevice(ctx->device) != hipSuccess || hipMalloc(&p, bytes + 64) != hipSuccess) return nullptr;
    return p;
}
void pqd_device_free(pqd_ctx* ctx, void* p) { if (ctx && p) { (void)hipSetDevice(ctx->device); (void)hipFree(p); } }

int pqd_copy_to_device(pqd_ctx* ctx, void* dst, const void* src, size_t bytes) {
    if (!ctx || !dst || !src) return PQD_E_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return PQD_OK;
}

int pqd_copy_to_host(pqd_ctx* ctx, void* dst, const void* src_device, size_t bytes) {
    if (!ctx || !dst || !src_device) return PQD_E_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemcpyAsync(dst, src_device, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return PQD_OK;
}

void pqd_plan_destroy(pqd_plan* pl) {
    if (!pl) return;
    (void)hipSetDevice(pl->ctx->device);
    (void)hipStreamSynchronize(pl->ctx->stream);
    pool_release(pl->ctx, pl->arena);      // the stream has been waited for: nothing uses the blocks any more
    pool_release(pl->ctx, pl->owned_src);
    for (auto& set : pl->ev)
        for (hipEvent_t e : set)
            if (e) (void)hipEventDestroy(e);
    if (pl->ev_fork) (void)hipEventDestroy(pl->ev_fork);
    if (pl->ev_join) (void)hipEventDestroy(pl->ev_join);
    if (pl->ev_vidx) (void)hipEventDestroy(pl->ev_vidx);
    if (pl->ev_join_copy) (void)hipEventDestroy(pl->ev_join_copy);
    if (pl->ev_levels) (void)hipEventDestroy(pl->ev_levels);
    if (pl->gexec) (void)hipGraphExecDestroy(pl->gexec);
    delete pl;
}

int pqd_plan_create(pqd_ctx* ctx, const pqd_chunk_desc* chunks, size_t n_chunks, const pqd_page_desc* pages,
                    size_t n_pages, const void* src_host_, size_t src_len, pqd_plan** out) {
    if (!ctx || !out || (!chunks && n_chunks) || (!pages && n_pages)) return PQD_E_ARG;
    if (n_chunks > 0x7fffffffu || n_pages > 0x7fffffffu) return fail(ctx, PQD_E_ARG, "too many chunks or pages");
    const u8* src_host = (const u8*)src_host_;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    pqd_plan* pl = new pqd_plan();
    pl->ctx = ctx;
    pl->src_len = src_len;
    struct Guard { pqd_plan* p; ~Guard() { if (p) pqd_plan_destroy(p); } } guard{pl};

    pl->chunks.resize(n_chunks);
    pl->pages.resize(n_pages);
    pl->chunk_width.resize(n_chunks);
    std::vector<u32> rep_bytes(n_pages, 0), def_bytes(n_pages, 0);
    std::vector<u64> lvl_off(n_pages, 0), val_stream_off(n_pages, 0), payload_off(n_pages, 0);
    std::vector<char> page_seen(n_pages, 0);
    std::vector<char> rep_on_host(n_pages, 0);  // the page's repetition level block was located through src_host: its bytes lie there

    // ---- validate descriptors; locate the level blocks and the values region of every page
    for (size_t c = 0; c < n_chunks; c++) {
        const pqd_chunk_desc& cd = chunks[c];
        ChunkRec& C = pl->chunks[c];
        memset(&C, 0, sizeof C);
        int w = 0;
        switch (cd.phys_type) {
        case PQD_INT32: case PQD_FLOAT: w = 4; break;
        case PQD_INT64: case PQD_DOUBLE: w = 8; break;
        case PQD_BOOLEAN: case PQD_BYTE_ARRAY: break;
        default: return fail(ctx, PQD_E_UNSUPPORTED, "unsupported physical type");
        }
        if (cd.codec != 0)
            return fail(ctx, PQD_E_UNSUPPORTED, "compressed chunk: its pages must be decoded first (pqd_decode does it for SNAPPY)");
        if (cd.max_def < 0 || cd.max_def > 255 || cd.max_rep < 0 || cd.max_rep > 255 || cd.max_rep > cd.max_def)
            return fail(ctx, PQD_E_UNSUPPORTED, "max levels out of range");
        if (cd.first_page < 0 || cd.n_pages < 0 || (size_t)cd.first_page + (size_t)cd.n_pages > n_pages || cd.num_values < 0)
            return fail(ctx, PQD_E_ARG, "chunk page range out of bounds");
        C.phys_type = cd.phys_type; C.max_def = cd.max_def; C.max_rep = cd.max_rep;
        C.width_bytes = (u32)w;
        C.first_page = (u32)cd.first_page; C.n_pages = (u32)cd.n_pages;
        pl->chunk_width[c] = w;
        u64 levels = 0, stream = 0;
        for (int k = 0; k < cd.n_pages; k++) {
            const size_t pi = (size_t)cd.first_page + k;
            if (page_seen[pi]) return fail(ctx, PQD_E_ARG, "page belongs to two chunks");
            page_seen[pi] = 1;
            const pqd_page_desc& pd = pages[pi];
            if (pd.src_off < 0 || pd.payload_len < 0 || pd.num_values < 0 ||
                (u64)pd.src_off + (u64)pd.payload_len > src_len)
                return fail(ctx, PQD_E_CORRUPT, "page payload outside the source buffer");
            u64 l = 0;
            for (int lv = 0; lv < 2; lv++) {
                const bool present = lv == 0 ? cd.max_rep > 0 : cd.max_def > 0;
                if (!present) continue;
                int32_t given = lv == 0 ? pd.rep_bytes : pd.def_bytes;
                if (given < 0) {
                    // int32 LE block length in front of the runs (rle.go:161-164)
                    if (!src_host) return fail(ctx, PQD_E_ARG, "level block sizes unknown and no host source given");
                    if (l + 4 > (u64)pd.payload_len) return fail(ctx, PQD_E_CORRUPT, "level block length outside the page");
                    int32_t len;
                    memcpy(&len, src_host + pd.src_off + l, 4);
                    if (len < 0) return fail(ctx, PQD_E_CORRUPT, "negative level block length");
                    given = (int32_t)((int64_t)len + 4 > 0x7fffffff ? 0x7fffffff : len + 4);
                    if (lv == 0) rep_on_host[pi] = 1;
                }
                if (given < 4 || l + (u64)given > (u64)pd.payload_len)
                    return fail(ctx, PQD_E_CORRUPT, "level block runs past the page payload");
                (lv == 0 ? rep_bytes[pi] : def_bytes[pi]) = (u32)given;
                l += (u64)given;
            }
            PageRec& P = pl->pages[pi];
            P.val_src_off = (u64)pd.src_off + l;
            P.val_len = (u32)((u64)pd.payload_len - l);
            P.num_values = (u32)pd.num_values;
            P.chunk = (u32)c;
            P.pad = 0;
            payload_off[pi] = (u64)pd.src_off;
            lvl_off[pi] = levels;
            if (cd.max_def > 0) pl->vidx_pages.push_back((u32)pi);
            val_stream_off[pi] = stream;
            levels += (u64)pd.num_values;
            stream += P.val_len;
            pl->bytes_in += pd.payload_len;
        }
        if (levels > 0x7ffffff0ull) return fail(ctx, PQD_E_UNSUPPORTED, "column chunk with 2^31 levels or more");  // int32 row offsets, 32-bit level indices
        C.n_levels = levels;
        C.stream_len = stream;
        C.req_values = (u64)cd.num_values;
        if (cd.max_def == 0) {
            // RequiredField.DoRead stops once sum(NumValues) >= pg.N (fields.go:118); the typed Read then
            // takes pg.N values.  For bools and strings the page split matters, so require an exact fit.
            if (levels < (u64)cd.num_values) return fail(ctx, PQD_E_CORRUPT, "pages hold fewer values than the chunk declares");
            if (w == 0 && levels != (u64)cd.num_values) return fail(ctx, PQD_E_NONCANONICAL, "page value counts exceed the chunk's");
        }
        if (bits_len((unsigned)cd.max_def) > 8) return fail(ctx, PQD_E_UNSUPPORTED, "level width > 8");
    }
    for (size_t p = 0; p < n_pages; p++)
        if (!page_seen[p]) return fail(ctx, PQD_E_ARG, "page not owned by any chunk");

    // Leaves under the same repeated groups carry the same repetition levels (Person.Friends.ID and .Age: one level
    // per friend in both), and a deterministic writer encodes them to the same bytes.  A chunk whose repetition level
    // blocks are byte for byte those of an earlier chunk of the batch (same pages, same value counts) shares that
    // chunk's decoded REPS, its row starts and its ROW_LEVEL_OFF instead of having them decoded and written again.
    // Only where the planner has the bytes at hand (blocks it located through src_host itself).
    std::vector<int> twin_of(n_chunks, -1);
    {
        const char* sh = getenv("PQD_SHARE_REPS");
        const bool share = !sh || atoi(sh) != 0;
        std::vector<std::pair<u64, size_t>> sigs;  // (signature, chunk) of the chunks that can be shared with
        for (size_t c = 0; c < n_chunks && share && src_host; c++) {
            const ChunkRec& C = pl->chunks[c];
            if (C.max_rep <= 0 || C.n_pages == 0) continue;
            bool ok = true;
            u64 sig = 0xcbf29ce484222325ull ^ ((u64)C.max_rep << 32) ^ C.n_pages;
            for (u32 k = 0; k < C.n_pages && ok; k++) {
                const size_t pi = C.first_page + k;
                ok = rep_on_host[pi] != 0;
                if (!ok) break;
                const u8* b = src_host + payload_off[pi];
                const u32 nb = rep_bytes[pi];
                sig = (sig ^ nb ^ ((u64)pl->pages[pi].num_values << 32)) * 0x100000001b3ull;
                u32 i = 0;
                for (; i + 8 <= nb; i += 8) { u64 x; memcpy(&x, b + i, 8); sig = (sig ^ x) * 0x9E3779B97F4A7C15ull; sig ^= sig >> 29; }
                for (; i < nb; i++) sig = (sig ^ b[i]) * 0x100000001b3ull;
            }
            if (!ok) continue;
            for (const auto& e : sigs) {
                if (e.first != sig) continue;
                const ChunkRec& D = pl->chunks[e.second];
                bool same = D.max_rep == C.max_rep && D.n_pages == C.n_pages && D.n_levels == C.n_levels;
                for (u32 k = 0; k < C.n_pages && same; k++) {
                    const size_t pa = C.first_page + k, pb = D.first_page + k;
                    same = rep_bytes[pa] == rep_bytes[pb] && pl->pages[pa].num_values == pl->pages[pb].num_values &&
                           memcmp(src_host + payload_off[pa], src_host + payload_off[pb], rep_bytes[pa]) == 0;
                }
                if (same) { twin_of[c] = (int)e.second; break; }
            }
            if (twin_of[c] < 0) sigs.emplace_back(sig, c);
        }
    }

    // ---- size the outputs and the scratch tables
    ArenaPlan ap;
    std::vector<size_t> off_out(n_chunks * N_OUT, (size_t)-1);
    pl->outs.resize(n_chunks);
    for (size_t c = 0; c < n_chunks; c++) {
        const ChunkRec& C = pl->chunks[c];
        size_t* o = &off_out[c * N_OUT];
        size_t* cap = pl->outs[c].cap;
        memset(cap, 0, sizeof pl->outs[c].cap);
        auto want = [&](int which, size_t bytes) { cap[which] = bytes; o[which] = ap.take(bytes + 64); };
        if (C.max_def > 0) {
            want(PQD_OUT_DEFS, C.n_levels);
            cap[PQD_OUT_VALIDITY] = (C.n_levels + 7) / 8;  // placed in the range that is zeroed before every run (below)
            want(PQD_OUT_VALUE_INDEX, ((C.n_levels + 63) / 64 + 1) * 4);
        }
        if (C.max_rep > 0) {
            if (twin_of[c] < 0) {
                want(PQD_OUT_REPS, C.n_levels);
                want(PQD_OUT_ROW_LEVEL_OFF, (C.n_levels + 1) * 4);
            } else {  // shared with the twin (patched below)
                cap[PQD_OUT_REPS] = C.n_levels;
                cap[PQD_OUT_ROW_LEVEL_OFF] = (C.n_levels + 1) * 4;
            }
            want(PQD_OUT_ROW_VALUE_OFF, (C.n_levels + 1) * 4);
        }
        if (C.width_bytes) want(PQD_OUT_VALUES, C.stream_len);
        else if (C.phys_type == PQD_BOOLEAN) want(PQD_OUT_VALUES, C.n_levels);
        else {
            want(PQD_OUT_OFFSETS, (C.n_levels + 1) * 4);
            want(PQD_OUT_CHARS, C.stream_len);
        }
        want(PQD_OUT_PAGE_NONNULL, (size_t)C.n_pages * 8);
    }

    // regions.  Super-tile size of the level decode (see pqd_device.h): the 8 KiB build when the batch is bound by the chain of
    // hand-overs inside its longest block, i.e. when all its super-tiles together are fewer than LV_CHAIN_RATIO times those
    // of that block; else the 4 KiB build
    {
        u64 n8 = 0, longest = 1;
        for (size_t pi = 0; pi < n_pages; pi++)
            for (int lv = 0; lv < 2; lv++) {
                const u32 nb = lv == 0 ? rep_bytes[pi] : def_bytes[pi];
                const u64 t = nb > 4 ? ((u64)nb - 4 + LV_T_SMALL_BATCH - 1) / LV_T_SMALL_BATCH : 1;
                if (nb) { n8 += t; longest = std::max(longest, t); }
            }
        pl->lv_tile = n8 > LV_CHAIN_RATIO * longest ? LV_T_LARGE_BATCH : LV_T_SMALL_BATCH;
        const char* force = getenv("PQD_LV_TILE");  // diagnostics: 4096 or 8192
        if (force && (atoi(force) == (int)LV_T_LARGE_BATCH || atoi(force) == (int)LV_T_SMALL_BATCH)) pl->lv_tile = (u32)atoi(force);
    }
    const u32 lv_tile = pl->lv_tile;
    u64 n_lv = 0;
    pl->stile_chain_prefix.push_back(0);
    pl->sgroup_prefix.push_back(0);
    pl->ssuper_prefix.push_back(0);
    pl->btile_prefix.push_back(0);
    pl->rtile_prefix.push_back(0);
    pl->seg_blk_prefix.push_back(0);
    auto push_prefix = [&](std::vector<u32>& v, u64 add, bool& ovf) {
        u64 nv = (u64)v.back() + add;
        if (nv > 0x7fffffffull) ovf = true;
        v.push_back((u32)nv);
    };
    bool ovf = false;
    u32 str_chunk_unit0 = 0;  // first unit (single string pass) of the BYTE_ARRAY chunk being planned
    for (size_t c = 0; c < n_chunks; c++) {
        const ChunkRec& C = pl->chunks[c];
        if (C.phys_type == PQD_BYTE_ARRAY) {
            pl->str_chunk_prefix.push_back((u32)pl->sregions.size());
            str_chunk_unit0 = (u32)pl->sunit_region.size();
        }
        for (u32 k = 0; k < C.n_pages; k++) {
            const u32 pi = C.first_page + k;
            const PageRec& P = pl->pages[pi];
            u64 l = 0;
            for (int lv = 0; lv < 2; lv++) {
                const u32 nb = lv == 0 ? rep_bytes[pi] : def_bytes[pi];
                if (!nb) continue;
                if (lv == 0 && twin_of[c] >= 0) { l += nb; continue; }  // decoded with the twin's
                Region R;
                memset(&R, 0, sizeof R);
                R.src_off = payload_off[pi] + l + 4;
                R.len = nb - 4;
                R.width = (u32)bits_len((unsigned)(lv == 0 ? C.max_rep : C.max_def));
                R.lv_base = (u32)n_lv;
                R.n_out = P.num_values;
                R.out = (u8*)(uintptr_t)lvl_off[pi];  // patched with the array base below
                R.page = pi;
                R.count_level = lv == 1 ? (u32)C.max_def : ERR_POS;
                if (lv == 0) {  // repetition levels: the decode also leaves the bitmap of rep == 0 for the row offsets
                    R.count_level = 0;
                    R.otile_base |= LV_BITMAP_ONLY;
                }
                // every block has at least one super-tile: an empty one still has to report its missing values
                const u32 n_sup = R.len ? (R.len + lv_tile - 1) / lv_tile : 1;
                for (u32 t = 0; t < n_sup; t++) pl->lv_order.push_back(LvTile{(u32)pl->lregions.size(), t});
                pl->lregions.push_back(R);
                n_lv += n_sup;
                if (n_lv > 0x7fffffffull) ovf = true;
                l += nb;
            }
            if (C.phys_type == PQD_BYTE_ARRAY) {
                Region R;
                memset(&R, 0, sizeof R);
                R.src_off = P.val_src_off;
                R.len = P.val_len;
                R.tile_base = pl->stile_chain_prefix.back();
                R.group_base = pl->sgroup_prefix.back();
                R.n_out = P.num_values;
                R.page = pi;
                R.count_level = ERR_POS;
                // single pass (k_str_fused): ranks and char positions are the chunk's, not the page's
                R.count_level = (u32)pl->sunit_region.size();  // first unit of the page
                R.lv_base = str_chunk_unit0;                    // first unit of the chunk
                R.run_base = val_stream_off[pi];
                pl->sunit_region.insert(pl->sunit_region.end(), ((size_t)P.val_len + SF_UNIT - 1) / SF_UNIT, (u32)pl->sregions.size());
                R.width = (u32)(C.n_levels < 0xFFFFFFFFull ? C.n_levels : 0xFFFFFFFFull);
                R.otile_base = (u32)pl->str_chunk_prefix.size() - 1;
                pl->sregions.push_back(R);
                push_prefix(pl->stile_chain_prefix, ((u64)P.val_len + CH_T - 1) / CH_T, ovf);
                push_prefix(pl->sgroup_prefix, ((u64)P.val_len + CH_G_STRINGS * CH_T - 1) / (CH_G_STRINGS * CH_T), ovf);
                push_prefix(pl->ssuper_prefix, ((u64)P.val_len + (u64)CH_SUPER * CH_G_STRINGS * CH_T - 1) / ((u64)CH_SUPER * CH_G_STRINGS * CH_T), ovf);
            } else if (C.phys_type == PQD_BOOLEAN) {
                pl->bool_pages.push_back(pi);
                push_prefix(pl->btile_prefix, ((u64)P.num_values + BT - 1) / BT, ovf);
            }
        }
        if (C.max_rep > 0) {  // per-row offsets: repeated columns only (otherwise row r is level r)
            pl->row_chunks.push_back((u32)c);
            push_prefix(pl->rtile_prefix, (C.n_levels + RT - 1) / RT, ovf);
        }
    }
    if (ovf) return fail(ctx, PQD_E_UNSUPPORTED, "batch too large for 32-bit tile indices");
    pl->str_chunk_prefix.push_back((u32)pl->sregions.size());
    pl->n_lvtiles = (u32)n_lv;
    // workgroups take super-tiles in this order: super-tile t of every block before super-tile t + 1 of any, so
    // the one a workgroup waits for was always taken earlier, and the chains of all blocks advance side by side
    std::stable_sort(pl->lv_order.begin(), pl->lv_order.end(), [](const LvTile& x, const LvTile& y) { return x.index < y.index; });
    const int lv_balance = getenv("PQD_LV_BALANCE") ? atoi(getenv("PQD_LV_BALANCE")) : 1;  // 0: off, 2: also for a few blocks (tests)
    if (lv_balance == 2 || (lv_balance == 1 && pl->lregions.size() >= 64)) {  // fewer blocks would leave XCDs without any
        // Workgroups are dealt to the eight XCDs in turn, so position p of the order runs on XCD p % 8.  Every block is given
        // to one XCD (its hand-overs then stay inside that XCD, which measured faster than blocks that go round the XCDs)
        // and the blocks are shared out so that every XCD gets the same number of super-tiles: in the order of the file a
        // batch of six level blocks per row group, three of them five times as long as the others, put 1.7 times the
        // super-tiles on the odd XCDs, and every other kernel's share of those XCDs ran late with them.  Longest blocks
        // first, each to the XCD with the least so far; positions that stay empty hold a workgroup that returns at once.
        constexpr u32 NX = 8;
        const u32 nreg = (u32)pl->lregions.size();
        std::vector<u32> n_sup(nreg, 0), xcd(nreg, 0), by_len(nreg);
        for (const LvTile& t : pl->lv_order) n_sup[t.region]++;
        for (u32 r = 0; r < nreg; r++) by_len[r] = r;
        std::stable_sort(by_len.begin(), by_len.end(), [&](u32 a, u32 b) { return n_sup[a] > n_sup[b]; });
        u64 load[NX] = {};
        for (u32 r : by_len) {
            u32 best = 0;
            for (u32 x = 1; x < NX; x++) if (load[x] < load[best]) best = x;
            xcd[r] = best;
            load[best] += n_sup[r];
        }
        std::vector<LvTile> q[NX];
        for (const LvTile& t : pl->lv_order) q[xcd[t.region]].push_back(t);  // keeps (index, region) order inside a queue
        size_t longest = 0;
        for (u32 x = 0; x < NX; x++) longest = std::max(longest, q[x].size());
        std::vector<LvTile> order(longest * NX, LvTile{ERR_POS, 0});
        for (u32 x = 0; x < NX; x++)
            for (size_t i = 0; i < q[x].size(); i++) order[i * NX + x] = q[x][i];
        while (!order.empty() && order.back().region == ERR_POS) order.pop_back();
        if (order.size() > 0x7fffffffull) return fail(ctx, PQD_E_UNSUPPORTED, "batch too large for 32-bit tile indices");
        pl->lv_order.swap(order);
    }
    pl->n_stiles_chain = pl->stile_chain_prefix.back();
    for (size_t r = 0; r < pl->sregions.size(); r++)
        pl->stile_region.insert(pl->stile_region.end(), pl->stile_chain_prefix[r + 1] - pl->stile_chain_prefix[r], (u32)r);
    pl->n_btiles = pl->btile_prefix.back();
    pl->n_rtiles = pl->rtile_prefix.back();
    for (size_t k = 0; k + 1 < pl->btile_prefix.size(); k++)
        pl->btile_owner.insert(pl->btile_owner.end(), pl->btile_prefix[k + 1] - pl->btile_prefix[k], (u32)k);
    for (size_t k = 0; k + 1 < pl->rtile_prefix.size(); k++)
        pl->rtile_owner.insert(pl->rtile_owner.end(), pl->rtile_prefix[k + 1] - pl->rtile_prefix[k], (u32)k);

    // static tables
    const size_t o_chunks = ap.take(n_chunks * sizeof(ChunkRec));
    const size_t o_pages = ap.take(n_pages * sizeof(PageRec));
    const size_t o_lreg = ap.take(pl->lregions.size() * sizeof(Region));
    const size_t o_sreg = ap.take(pl->sregions.size() * sizeof(Region));
    const size_t o_lvo = ap.take(pl->lv_order.size() * sizeof(LvTile));
    pl->page_lvl_off.assign(lvl_off.begin(), lvl_off.end());
    const size_t o_vip = ap.take(pl->vidx_pages.size() * 4), o_plo = ap.take(n_pages * 8);
    const size_t o_sctp = ap.take(pl->stile_chain_prefix.size() * 4);
    const size_t o_str = ap.take(pl->stile_region.size() * 4);
    {   // Launch order of the units: the k-th unit of every chunk before the k+1-th of any, so that the units in flight are
        // spread over all chunks and a unit's predecessors in its chunk are few and old; and the chunks of one position
        // rotated by the position, so that a chunk's units go round the XCDs (workgroups are dealt to the XCDs in turn: with
        // eight chunks in a fixed order every XCD held the units of one chunk only, and stood still whenever that chunk's
        // ranks did: 2.3 against 3.4 TB/s on eight chunks of optional strings, 3.2 against 3.9 on 32; PQD_STR_ROTATE=0
        // restores the fixed order for comparison).  A unit's predecessor in its chunk still comes before it.
        const size_t nu = pl->sunit_region.size();
        pl->sunits.resize(nu);
        std::vector<std::pair<u64, u32>> key(nu);
        const bool rot = !(getenv("PQD_STR_ROTATE") && atoi(getenv("PQD_STR_ROTATE")) == 0);
        const u32 nch = pl->str_chunk_prefix.size() > 1 ? (u32)pl->str_chunk_prefix.size() - 1 : 1;
        for (size_t u = 0; u < nu; u++) {
            const Region& R = pl->sregions[pl->sunit_region[u]];
            const u32 k = (u32)u - R.lv_base;
            key[u] = {(u64)k << 32 | (rot ? (R.otile_base + k) % nch : R.otile_base), (u32)u};
        }
        std::sort(key.begin(), key.end());
        for (size_t k = 0; k < nu; k++) {
            const u32 u = key[k].second, r = pl->sunit_region[u];
            const Region& R = pl->sregions[r];
            const u32 tb0 = (u - R.count_level) * SF_UNIT;
            pl->sunits[k] = StrUnit{R.src_off + tb0, tb0, R.len - tb0, r, u};
        }
    }
    const size_t o_suo = ap.take(pl->sunits.size() * sizeof(StrUnit));
    const size_t o_sgp = ap.take(pl->sgroup_prefix.size() * 4);
    const size_t o_ssp = ap.take(pl->ssuper_prefix.size() * 4);
    const size_t o_bp = ap.take(pl->bool_pages.size() * 4);
    const size_t o_btp = ap.take(pl->btile_prefix.size() * 4);
    const size_t o_rc = ap.take(pl->row_chunks.size() * 4);
    const size_t o_rtp = ap.take(pl->rtile_prefix.size() * 4);
    const size_t o_bto = ap.take(pl->btile_owner.size() * 4), o_rto = ap.take(pl->rtile_owner.size() * 4);
    // scratch
    const size_t o_sexit = ap.take((size_t)pl->n_stiles_chain * CH_E_STRINGS * sizeof(StrExit));
    const size_t o_scp = ap.take((size_t)pl->n_stiles_chain * CH_E_STRINGS * 16);
    const size_t o_scand = ap.take((size_t)pl->n_stiles_chain * CH_E_STRINGS * 2);
    const size_t o_sncand = ap.take((size_t)pl->n_stiles_chain * 2), o_sgpre = ap.take((size_t)pl->n_stiles_chain * 2);
    const size_t o_sspre = ap.take((size_t)pl->n_stiles_chain * 2);
    const u32 n_sgroups = pl->sgroup_prefix.back();
    const size_t o_sgexit = ap.take((size_t)n_sgroups * CH_E_STRINGS * sizeof(ExitRec));
    const u32 n_ssupers = pl->ssuper_prefix.back();
    const size_t o_ssexit = ap.take((size_t)n_ssupers * CH_E_STRINGS * sizeof(ExitRec));
    const size_t o_stot = ap.take(pl->sregions.size() * sizeof(RegionTotal));
    const size_t o_scpf = ap.take(pl->str_chunk_prefix.size() * 4), o_spd = ap.take(pl->sregions.size() * sizeof(PageDyn));
    const size_t o_rtiles = ap.take((size_t)pl->n_rtiles * sizeof(RowTile));
    const size_t o_cdyn = ap.take(n_chunks * sizeof(ChunkDyn));
    const bool lv_stamps = getenv("PQD_LV_STAMPS") != nullptr;  // diagnostics: per-workgroup phase stamps of k_levels
    const size_t o_lvst = lv_stamps ? ap.take(pl->lv_order.size() * LV_STAMPS * 8) : 0;  // a record per position of the order
    const bool str_stamps = getenv("PQD_STR_STAMPS") != nullptr;  // diagnostics: per-workgroup phase stamps of k_str_fused
    const size_t o_sfst = str_stamps ? ap.take(pl->sunit_region.size() * SF_STAMPS * 8) : 0;
    // reset to 0xFF every run
    const size_t o_ff = ap.total;
    const size_t o_ste = ap.take((size_t)pl->n_stiles_chain * sizeof(TileEntry));
    const size_t o_sge = ap.take((size_t)n_sgroups * sizeof(TileEntry));
    const size_t o_sse = ap.take((size_t)n_ssupers * sizeof(TileEntry));
    const size_t ff_end = ap.total;
    // reset to 0 every run
    const size_t o_zero = ap.total;
    const size_t o_pdyn = ap.take(n_pages * sizeof(PageDyn));
    // validity bitmaps: the level kernel ORs the words that two workgroups share
    for (size_t c = 0; c < n_chunks; c++)
        if (pl->chunks[c].max_def > 0) off_out[c * N_OUT + PQD_OUT_VALIDITY] = ap.take(pl->outs[c].cap[PQD_OUT_VALIDITY] + 64);
    std::vector<size_t> off_rstarts(n_chunks, (size_t)-1);  // row-start bitmaps of the repeated chunks, same treatment
    for (size_t c = 0; c < n_chunks; c++)
        if (pl->chunks[c].max_rep > 0 && twin_of[c] < 0) off_rstarts[c] = ap.take((pl->chunks[c].n_levels + 7) / 8 + 64);
    const size_t o_status = ap.take(16);
    const size_t o_lvticket = ap.take(16);
    const size_t o_lvpos = ap.take((size_t)pl->n_lvtiles * 8), o_lvval = ap.take((size_t)pl->n_lvtiles * 8);  // hand-off granules
    // strings, single pass: a status word per tile, a flag per chunk, one flag for the batch
    const size_t o_sstat = ap.take(pl->sunit_region.size() * 8);
    const size_t o_sfail = ap.take(pl->str_chunk_prefix.size() * 4), o_srep = ap.take(pl->str_chunk_prefix.size() * 4), o_sany = ap.take(16);
    const size_t zero_end = ap.total;
    // the copy segment tables depend on output addresses → placed last, sized now
    size_t n_segs = 0;
    u64 max_blocks = 0;  // upper bound of the gather's blocks: a segment's destination may be misaligned by up to 15 bytes
    for (size_t c = 0; c < n_chunks; c++)
        if (pl->chunks[c].width_bytes)
            for (u32 k = 0; k < pl->chunks[c].n_pages; k++) {
                const u32 vl = pl->pages[pl->chunks[c].first_page + k].val_len;
                n_segs += vl > 0;
                if (vl) max_blocks += ((u64)vl + 15 + CP_BLK - 1) / CP_BLK;
            }
    const size_t o_segs = ap.take(n_segs * sizeof(Seg));
    const size_t o_segp = ap.take((n_segs + 1) * 4);
    const size_t o_segg = ap.take((size_t)(max_blocks / CP_GRP + 2) * 4);  // segment of every CP_GRP-th block

    pl->arena_bytes = ap.total;
    pl->arena = (u8*)pool_alloc(ctx, pl->arena_bytes);
    if (!pl->arena) return fail(ctx, PQD_E_NOMEM, "hipMalloc of the plan arena failed");
    u8* A = pl->arena;

    // patch pointers
    for (size_t c = 0; c < n_chunks; c++) {
        ChunkRec& C = pl->chunks[c];
        u8** ptrs[N_OUT] = {&C.defs, &C.reps, &C.values, &C.offsets, &C.chars, &C.validity, &C.page_nonnull,
                            &C.row_level_off, &C.row_value_off, &C.value_index};
        for (int w = 0; w < N_OUT; w++) {
            const size_t o = off_out[c * N_OUT + w];
            *ptrs[w] = o == (size_t)-1 ? nullptr : A + o;
            pl->outs[c].p[w] = *ptrs[w];
        }
        C.row_starts = off_rstarts[c] == (size_t)-1 ? nullptr : A + off_rstarts[c];
    }
    for (size_t c = 0; c < n_chunks; c++) {
        if (twin_of[c] < 0) continue;
        ChunkRec& C = pl->chunks[c];
        const ChunkRec& T = pl->chunks[(size_t)twin_of[c]];  // an earlier chunk, never a twin itself
        C.reps = T.reps; C.row_starts = T.row_starts; C.row_level_off = T.row_level_off;
        C.shares_rep = 1;
        pl->outs[c].p[PQD_OUT_REPS] = T.reps;
        pl->outs[c].p[PQD_OUT_ROW_LEVEL_OFF] = T.row_level_off;
        pl->n_shared_rep++;
    }
    for (Region& R : pl->lregions) {
        const ChunkRec& C = pl->chunks[pl->pages[R.page].chunk];
        const u64 lo = (u64)(uintptr_t)R.out;
        const bool is_rep = (R.otile_base & LV_BITMAP_ONLY) != 0;
        R.out = (is_rep ? C.reps : C.defs) + lo;
        R.run_base = lo;                                // level index of the page's first level in its chunk
        R.out2 = is_rep ? C.row_starts : C.validity;    // the levels also yield a bitmap: rep == 0 / def == max_def
    }
    for (Region& R : pl->sregions) {
        R.out = pl->chunks[pl->pages[R.page].chunk].offsets;
        R.out2 = pl->chunks[pl->pages[R.page].chunk].chars;
    }

    // PLAIN value segments: the concatenation of the pages' values regions (fields.go:131,313)
    for (size_t c = 0; c < n_chunks; c++) {
        const ChunkRec& C = pl->chunks[c];
        if (!C.width_bytes) continue;
        for (u32 k = 0; k < C.n_pages; k++) {
            const u32 pi = C.first_page + k;
            const PageRec& P = pl->pages[pi];
            if (!P.val_len) continue;
            Seg s;
            s.src_off = P.val_src_off;
            s.dst = C.values + val_stream_off[pi];
            s.len = P.val_len;
            pl->segs.push_back(s);
            const u64 a = (u64)(uintptr_t)s.dst & 15;
            bool o2 = false;
            push_prefix(pl->seg_blk_prefix, (a + s.len + CP_BLK - 1) / CP_BLK, o2);
            if (o2) return fail(ctx, PQD_E_UNSUPPORTED, "batch too large for 32-bit block indices");
            pl->copy_bytes += 2 * (int64_t)s.len;
        }
    }
    pl->n_copy_blocks = pl->seg_blk_prefix.back();
    {   // where the owner search of a block starts: the segment that holds block g * CP_GRP
        u32 sidx = 0;
        for (u64 b = 0; b < pl->n_copy_blocks; b += CP_GRP) {
            while (pl->seg_blk_prefix[sidx + 1] <= b) sidx++;
            pl->seg_grp.push_back(sidx);
        }
        pl->seg_grp.push_back(pl->segs.empty() ? 0u : (u32)pl->segs.size() - 1);  // closes the last group
        if (pl->seg_grp.size() > max_blocks / CP_GRP + 2) return fail(ctx, PQD_E_HIP, "copy block table larger than planned");
    }

    pl->d_chunks = (ChunkRec*)(A + o_chunks); pl->d_pages = (PageRec*)(A + o_pages);
    pl->d_lregions = (Region*)(A + o_lreg); pl->d_sregions = (Region*)(A + o_sreg);
    pl->d_stile_chain_prefix = (u32*)(A + o_sctp);
    pl->d_sgroup_prefix = (u32*)(A + o_sgp);
    pl->d_ssuper_prefix = (u32*)(A + o_ssp);
    pl->d_sgexit = (ExitRec*)(A + o_sgexit);
    pl->d_sgroup_entry = (TileEntry*)(A + o_sge);
    pl->d_bool_pages = (u32*)(A + o_bp); pl->d_btile_prefix = (u32*)(A + o_btp);
    pl->d_row_chunks = (u32*)(A + o_rc); pl->d_rtile_prefix = (u32*)(A + o_rtp);
    pl->d_btile_owner = (u32*)(A + o_bto); pl->d_rtile_owner = (u32*)(A + o_rto);
    pl->d_sexit = (ExitRec*)(A + o_sexit);
    pl->d_stotals = (RegionTotal*)(A + o_stot);
    pl->d_str_chunk_prefix = (u32*)(A + o_scpf); pl->d_str_pd = (PageDyn*)(A + o_spd);
    pl->d_row_tiles = (RowTile*)(A + o_rtiles);
    pl->d_chunk_dyn = (ChunkDyn*)(A + o_cdyn);
    pl->d_stile_entry = (TileEntry*)(A + o_ste);
    pl->d_page_dyn = (PageDyn*)(A + o_pdyn); pl->d_status = (int*)(A + o_status);
    pl->d_segs = (Seg*)(A + o_segs); pl->d_seg_blk_prefix = (u32*)(A + o_segp); pl->d_seg_grp = (u32*)(A + o_segg);
    pl->ff_begin = A + o_ff; pl->ff_bytes = ff_end - o_ff;
    pl->zero_begin = A + o_zero; pl->zero_bytes = zero_end - o_zero;

    pl->lv = LvArgs{pl->d_lregions, (const LvTile*)(A + o_lvo), (u32)pl->lv_order.size(), (u64*)(A + o_lvpos), (u64*)(A + o_lvval),
                    (u32*)(A + o_lvticket), pl->d_page_dyn, pl->d_status, lv_stamps ? (u64*)(A + o_lvst) : nullptr};
    pl->schain = ChainArgs{pl->d_sregions, pl->d_stile_chain_prefix, pl->d_sgroup_prefix, (u32)pl->sregions.size(), pl->n_stiles_chain,
                           n_sgroups, pl->d_sexit, pl->d_sgexit, pl->d_stile_entry, pl->d_sgroup_entry, pl->d_stotals, A + o_scp,
                           (u32*)(A + o_ssp), n_ssupers, (ExitRec*)(A + o_ssexit), (TileEntry*)(A + o_sse)};
    pl->schain.tile_region = (u32*)(A + o_str);
    if (ctx->str_fused) {
        pl->schain.str_stat = (u64*)(A + o_sstat);
        pl->schain.str_failed = (u32*)(A + o_sfail);
        pl->schain.str_repairs = (u32*)(A + o_srep);
        pl->schain.str_any = (u32*)(A + o_sany);
        pl->schain.n_units = (u32)pl->sunit_region.size();
        pl->schain.units = (StrUnit*)(A + o_suo);
        pl->schain.str_stamps = str_stamps ? (u64*)(A + o_sfst) : nullptr;
    }
    pl->schain.str_index = StrIndex{(const unsigned short*)(A + o_scand),
                                    (const unsigned short*)(A + o_sncand), (unsigned short*)(A + o_sgpre), (unsigned short*)(A + o_sspre)};

    int rc;
    if ((rc = upload(ctx, pl->d_chunks, pl->chunks))) return rc;
    if ((rc = upload(ctx, pl->d_pages, pl->pages))) return rc;
    if ((rc = upload(ctx, pl->d_lregions, pl->lregions))) return rc;
    if ((rc = upload(ctx, pl->d_sregions, pl->sregions))) return rc;
    if ((rc = upload(ctx, pl->d_str_chunk_prefix, pl->str_chunk_prefix))) return rc;
    if ((rc = upload(ctx, (LvTile*)(A + o_lvo), pl->lv_order))) return rc;
    pl->d_vidx_pages = (u32*)(A + o_vip); pl->d_lvl_off = (u64*)(A + o_plo);
    if ((rc = upload(ctx, pl->d_vidx_pages, pl->vidx_pages))) return rc;
    if ((rc = upload(ctx, pl->d_lvl_off, pl->page_lvl_off))) return rc;
    if ((rc = upload(ctx, pl->d_stile_chain_prefix, pl->stile_chain_prefix))) return rc;
    if ((rc = upload(ctx, (u32*)(A + o_str), pl->stile_region))) return rc;
    if ((rc = upload(ctx, (StrUnit*)(A + o_suo), pl->sunits))) return rc;
    if ((rc = upload(ctx, pl->d_sgroup_prefix, pl->sgroup_prefix))) return rc;
    if ((rc = upload(ctx, pl->d_ssuper_prefix, pl->ssuper_prefix))) return rc;
    if ((rc = upload(ctx, pl->d_bool_pages, pl->bool_pages))) return rc;
    if ((rc = upload(ctx, pl->d_btile_prefix, pl->btile_prefix))) return rc;
    if ((rc = upload(ctx, pl->d_row_chunks, pl->row_chunks))) return rc;
    if ((rc = upload(ctx, pl->d_rtile_prefix, pl->rtile_prefix))) return rc;
    if ((rc = upload(ctx, pl->d_btile_owner, pl->btile_owner))) return rc;
    if ((rc = upload(ctx, pl->d_rtile_owner, pl->rtile_owner))) return rc;
    if ((rc = upload(ctx, pl->d_segs, pl->segs))) return rc;
    if ((rc = upload(ctx, pl->d_seg_blk_prefix, pl->seg_blk_prefix))) return rc;
    if ((rc = upload(ctx, pl->d_seg_grp, pl->seg_grp))) return rc;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (auto& set : pl->ev)
        for (hipEvent_t& e : set) HIP_TRY(ctx, hipEventCreate(&e));
    HIP_TRY(ctx, hipEventCreateWithFlags(&pl->ev_fork, hipEventDisableTiming));
    HIP_TRY(ctx, hipEventCreateWithFlags(&pl->ev_join, hipEventDisableTiming));
    HIP_TRY(ctx, hipEventCreateWithFlags(&pl->ev_vidx, hipEventDisableTiming));
    HIP_TRY(ctx, hipEventCreateWithFlags(&pl->ev_join_copy, hipEventDisableTiming));
    HIP_TRY(ctx, hipEventCreateWithFlags(&pl->ev_levels, hipEventDisableTiming));
    pl->h_chunk_dyn.resize(n_chunks);
    guard.p = nullptr;
    *out = pl;
    return PQD_OK;
}

// Queues one run of the plan on the decoder's streams.  ev_c0 / ev_c1 bracket the PLAIN value gather when
// given; under graph capture they are absent and a plain join event orders the two streams.
static int enqueue_run(pqd_plan* pl, const u8* src, hipEvent_t ev_c0, hipEvent_t ev_c1, bool one_stream = false) {
    pqd_ctx* ctx = pl->ctx;
    hipStream_t st = ctx->stream;
    const u32 n_chunks = (u32)pl->chunks.size(), n_pages = (u32)pl->pages.size();
    const u32 n_lreg = (u32)pl->lregions.size(), n_sreg = (u32)pl->sregions.size();
    if (pl->n_stiles_chain && !pl->schain.str_stat)  // behind the single pass the tables are reset on the device, and only when needed
        HIP_TRY(ctx, hipMemsetAsync(pl->ff_begin, 0xFF, pl->ff_bytes, st));
    HIP_TRY(ctx, hipMemsetAsync(pl->zero_begin, 0, pl->zero_bytes, st));
    if (ctx->str_giveup && pl->schain.str_failed) {
        HIP_TRY(ctx, hipMemsetAsync(pl->schain.str_failed, 1, (pl->str_chunk_prefix.size() - 1) * 4, st));
        HIP_TRY(ctx, hipMemsetAsync(pl->schain.str_any, 1, 4, st));
    }

    // Three streams.  The PLAIN value gather is independent of the levels (the reference reads the concatenated
    // values regions as one stream) and HBM-bound, while the level and string kernels are instruction-bound, so
    // the gather runs beside them on a stream of its own.  The strings (exit tables, hops, totals, marking pass)
    // need nothing from the levels either and go to another side stream; only the final comparison of their
    // counts with the levels' waits.  Levels, page scan, bools and row offsets stay on the main stream.
    const bool overlap = ctx->overlap && !one_stream;
    const bool str_early = overlap && n_lreg && n_sreg;
    const bool copy_side = overlap && pl->n_copy_blocks && (n_lreg || n_sreg);
    hipStream_t cst = copy_side ? ctx->stream3 : st, sst = str_early ? ctx->stream2 : st;
    // diagnostics (PQD_LEVELS_FIRST=1): the level decode alone, the other streams start behind it
    if (ctx->levels_first && n_lreg) HIP_TRY(ctx, launch_levels(st, src, pl->lv, pl->lv_tile));
    if (copy_side || str_early) HIP_TRY(ctx, hipEventRecord(pl->ev_fork, st));
    if (copy_side) HIP_TRY(ctx, hipStreamWaitEvent(cst, pl->ev_fork, 0));
    if (ev_c0) HIP_TRY(ctx, hipEventRecord(ev_c0, cst));
    if (!getenv("PQD_SKIP_COPY"))  // diagnostics: what the step takes without the gather
    HIP_TRY(ctx, launch_copy(cst, src, pl->d_segs, pl->d_seg_blk_prefix, pl->d_seg_grp, (u32)pl->segs.size(), pl->n_copy_blocks, ctx->copy_grid));
    if (ev_c1) HIP_TRY(ctx, hipEventRecord(ev_c1, cst));
    if (copy_side) HIP_TRY(ctx, hipEventRecord(pl->ev_join_copy, cst));
    // strings, both passes: a page's place in the chunk's offsets and chars follows from the string counts of the
    // first pass (k_str_page_scan), so nothing here waits for the levels; k_str_check compares with them at the end
    if (n_sreg) {
        if (str_early) HIP_TRY(ctx, hipStreamWaitEvent(sst, pl->ev_fork, 0));
        // one pass over the page bytes; the two passes behind it only touch the chunks it gave up
        if (pl->schain.str_stat) HIP_TRY(ctx, launch_str_fused(sst, src, pl->schain, pl->d_str_pd, pl->ff_begin, pl->ff_bytes));
        HIP_TRY(ctx, launch_chain(sst, KIND_STRINGS, src, pl->schain, nullptr, nullptr, pl->d_str_pd, pl->d_status, 0));
        HIP_TRY(ctx, launch_str_page_scan(sst, pl->d_sregions, pl->d_str_chunk_prefix, (u32)pl->str_chunk_prefix.size() - 1,
                                          pl->d_stotals, pl->d_str_pd, pl->schain.str_failed));
    }

    // levels, then everything that only needs them
    if (n_lreg && !ctx->levels_first) HIP_TRY(ctx, launch_levels(st, src, pl->lv, pl->lv_tile));
    HIP_TRY(ctx, launch_page_scan(st, pl->d_chunks, n_chunks, pl->d_pages, n_pages, pl->d_page_dyn, pl->d_chunk_dyn, pl->d_status));
    if (n_sreg) {
        if (str_early && ctx->str_after_levels) {  // diagnostics: the marking pass behind the level decode, as it was when it took the page offsets from there
            HIP_TRY(ctx, hipEventRecord(pl->ev_levels, st));
            HIP_TRY(ctx, hipStreamWaitEvent(sst, pl->ev_levels, 0));
        }
        HIP_TRY(ctx, launch_chain(sst, KIND_STRINGS, src, pl->schain, nullptr, nullptr, pl->d_str_pd, pl->d_status, 1));
        if (str_early) HIP_TRY(ctx, hipEventRecord(pl->ev_join, sst));
    }
    // The value index needs the page scan only.  It runs beside the bools and the row offsets on a stream that is idle by
    // then: the strings' in a batch without strings (nested columns: three launches shorter on the main stream), else the
    // gather's, which ends long before the level decode, when the main stream is the longer one (it saves the main stream
    // a launch and costs the step a join: wide columns 1.483 -> 1.44 ms, the batch of all column kinds, where the string
    // pass ends last either way, 3.395 -> 3.401 ms).  The two streams' lengths are estimated from the kernels' rates
    // alone: 21 ns per 4 KiB of level bytes, 3.5 ns per row tile, 7.7 ns per string unit.  PQD_VIDX_SIDE=0 / 1 decides
    // for every batch with strings instead.
    static const int vidx_env = getenv("PQD_VIDX_SIDE") ? atoi(getenv("PQD_VIDX_SIDE")) : -1;
    const double main_ns = 21.0 * pl->n_lvtiles * (pl->lv_tile / 4096.0) + 3.5 * pl->n_rtiles, str_ns = 7.7 * pl->schain.n_units;
    const bool vidx_with_strings = vidx_env >= 0 ? vidx_env != 0 : main_ns > str_ns;
    const bool vidx_side = overlap && (!n_sreg || vidx_with_strings) && n_lreg && !pl->row_chunks.empty();
    hipStream_t vst = n_sreg ? ctx->stream3 : ctx->stream2;
    if (vidx_side) {
        HIP_TRY(ctx, hipEventRecord(pl->ev_levels, st));
        HIP_TRY(ctx, hipStreamWaitEvent(vst, pl->ev_levels, 0));
    }
    // the row offsets need the bitmaps of the level decode and nothing else: the longest kernel of this stream goes first
    // (the value index beside it when the strings' stream is idle, else behind it)
    if (ctx->rows_first && !vidx_side)
        HIP_TRY(ctx, launch_rows(st, pl->d_chunks, pl->d_rtile_prefix, pl->d_rtile_owner, pl->d_row_chunks, (u32)pl->row_chunks.size(), pl->n_rtiles,
                                 pl->d_row_tiles, pl->d_chunk_dyn));
    HIP_TRY(ctx, launch_value_index(vidx_side ? vst : st, pl->d_chunks, pl->d_pages, pl->d_page_dyn, pl->d_vidx_pages,
                                    (u32)pl->vidx_pages.size(), pl->d_lvl_off));
    if (vidx_side) HIP_TRY(ctx, hipEventRecord(pl->ev_vidx, vst));
    HIP_TRY(ctx, launch_bools(st, src, pl->d_pages, pl->d_bool_pages, pl->d_btile_prefix, pl->d_btile_owner, (u32)pl->bool_pages.size(), pl->n_btiles,
                              pl->d_chunks, pl->d_page_dyn, pl->d_status));
    if (!ctx->rows_first || vidx_side)
        HIP_TRY(ctx, launch_rows(st, pl->d_chunks, pl->d_rtile_prefix, pl->d_rtile_owner, pl->d_row_chunks, (u32)pl->row_chunks.size(), pl->n_rtiles,
                                 pl->d_row_tiles, pl->d_chunk_dyn));
    if (n_sreg) {
        if (str_early) HIP_TRY(ctx, hipStreamWaitEvent(st, pl->ev_join, 0));
        HIP_TRY(ctx, launch_str_check(st, pl->d_pages, pl->d_chunks, pl->d_page_dyn, pl->d_sregions, n_sreg, pl->d_stotals, pl->d_status));
    }
    if (vidx_side) HIP_TRY(ctx, hipStreamWaitEvent(st, pl->ev_vidx, 0));
    if (copy_side) HIP_TRY(ctx, hipStreamWaitEvent(st, pl->ev_join_copy, 0));
    return PQD_OK;
}

int pqd_plan_run(pqd_plan* pl, const void* src_device) {
    if (!pl || !src_device) return PQD_E_ARG;
    pqd_ctx* ctx = pl->ctx;
    hipStream_t st = ctx->stream;
    const u8* src = (const u8*)src_device;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    pl->last_src = src;
    pl->synced = false;
    hipEvent_t* ev = pl->ev[pl->runs % pqd_plan::EV_RING];
    HIP_TRY(ctx, hipEventRecord(ev[0], st));
    // Batches with level or string chains are a dozen or more short launches: those are captured once into a
    // HIP graph and replayed (no per-gather timing in that mode).  A pure PLAIN batch is one long kernel.
    const bool use_graph = ctx->graph && (!pl->lregions.empty() || !pl->sregions.empty());
    if (use_graph) {
        if (!pl->gexec || pl->gsrc != src) {
            if (pl->gexec) { (void)hipGraphExecDestroy(pl->gexec); pl->gexec = nullptr; }
            HIP_TRY(ctx, hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            // One stream under capture: a graph with parallel branches runs them on streams of the runtime's own, on top of
            // this context's three, and more streams than hardware queues (four per process here) share queues: replays of
            // such a graph were seen to take seconds (6-110 s for three replays of a 100 us batch, no device error, fast
            // again under a profiler that serialises kernels).  A linear graph has no such branches.
            const int rc = enqueue_run(pl, src, nullptr, nullptr, true);
            hipGraph_t g = nullptr;
            const hipError_t e = hipStreamEndCapture(st, &g);
            if (rc) { if (g) (void)hipGraphDestroy(g); return rc; }
            if (e != hipSuccess) return fail(ctx, PQD_E_HIP, "hipStreamEndCapture", e);
            const hipError_t e2 = hipGraphInstantiate(&pl->gexec, g, nullptr, nullptr, 0);
            (void)hipGraphDestroy(g);
            if (e2 != hipSuccess) { pl->gexec = nullptr; return fail(ctx, PQD_E_HIP, "hipGraphInstantiate", e2); }
            pl->gsrc = src;
        }
        HIP_TRY(ctx, hipEventRecord(ev[2], st));
        HIP_TRY(ctx, hipEventRecord(ev[3], st));
        HIP_TRY(ctx, hipGraphLaunch(pl->gexec, st));
    } else {
        const int rc = enqueue_run(pl, src, ev[2], ev[3]);
        if (rc) return rc;
    }
    HIP_TRY(ctx, hipEventRecord(ev[1], st));
    pl->runs++;
    pl->ran = true;
    return PQD_OK;
}

int pqd_plan_sync(pqd_plan* pl) {
    if (!pl || !pl->ran) return PQD_E_ARG;
    pqd_ctx* ctx = pl->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (!pl->h_chunk_dyn.empty())
        HIP_TRY(ctx, hipMemcpyAsync(pl->h_chunk_dyn.data(), pl->d_chunk_dyn, pl->h_chunk_dyn.size() * sizeof(ChunkDyn),
                                    hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(pl->h_status, pl->d_status, 16, hipMemcpyDeviceToHost, ctx->stream));
    pl->h_str_failed.assign(pl->schain.str_failed ? pl->str_chunk_prefix.size() - 1 : 0, 0);
    if (!pl->h_str_failed.empty())
        HIP_TRY(ctx, hipMemcpyAsync(pl->h_str_failed.data(), pl->schain.str_failed, pl->h_str_failed.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    pl->synced = true;
    pl->last_status = map_status(pl->h_status[0]);
    if (pl->last_status) {
        char b[128];
        if (pl->h_status[0] == ST_INTERNAL)
            snprintf(b, sizeof b, "decode failed: a workgroup of block %d gave up waiting for tile %d (%s)", pl->h_status[1], pl->h_status[2],
                     pl->h_status[3] == 0 ? "position" : pl->h_status[3] == 1 ? "value count" : "tile flag");
        else
            snprintf(b, sizeof b, "decode failed: device status %d at item %d", pl->h_status[0], pl->h_status[1]);
        ctx->err = b;
    }
    return pl->last_status;
}

int pqd_plan_run_ms(pqd_plan* pl, int runs_back, float* total_ms, float* plain_copy_ms) {
    if (!pl || !pl->synced || runs_back < 0 || (uint64_t)runs_back >= pl->runs || runs_back >= pqd_plan::EV_RING)
        return PQD_E_ARG;
    hipEvent_t* ev = pl->ev[(pl->runs - 1 - (uint64_t)runs_back) % pqd_plan::EV_RING];
    if (total_ms) HIP_TRY(pl->ctx, hipEventElapsedTime(total_ms, ev[0], ev[1]));
    if (plain_copy_ms) HIP_TRY(pl->ctx, hipEventElapsedTime(plain_copy_ms, ev[2], ev[3]));
    return PQD_OK;
}

int pqd_plan_last_ms(pqd_plan* pl, float* total_ms, float* plain_copy_ms) {
    return pqd_plan_run_ms(pl, 0, total_ms, plain_copy_ms);
}

int pqd_plan_chunk_info(pqd_plan* pl, size_t c, pqd_chunk_info* info) {
    if (!pl || !info || c >= pl->chunks.size() || !pl->synced) return PQD_E_ARG;
    const ChunkRec& C = pl->chunks[c];
    const ChunkDyn& D = pl->h_chunk_dyn[c];
    info->n_levels = C.max_def > 0 ? (int64_t)C.n_levels : 0;
    info->n_values = (int64_t)D.n_values;
    info->value_bytes = C.width_bytes ? (int64_t)D.n_values * C.width_bytes : (C.phys_type == PQD_BOOLEAN ? (int64_t)D.n_values : 0);
    info->chars_bytes = C.phys_type == PQD_BYTE_ARRAY ? (int64_t)D.chars : 0;
    info->n_rows = C.max_def > 0 ? (int64_t)D.n_rows : (int64_t)D.n_values;
    info->n_pages = C.n_pages;
    return PQD_OK;
}

int pqd_plan_bytes(pqd_plan* pl, int64_t* bytes_in, int64_t* bytes_out, int64_t* plain_copy_bytes) {
    if (!pl) return PQD_E_ARG;
    if (bytes_in) *bytes_in = pl->bytes_in;
    if (plain_copy_bytes) *plain_copy_bytes = pl->copy_bytes;
    if (bytes_out) {
        if (!pl->synced) return PQD_E_ARG;
        int64_t o = 0;
        for (size_t c = 0; c < pl->chunks.size(); c++) {
            const ChunkRec& C = pl->chunks[c];
            const ChunkDyn& D = pl->h_chunk_dyn[c];
            if (C.max_def > 0) o += (int64_t)C.n_levels;
            if (C.max_rep > 0) o += (int64_t)C.n_levels;
            if (C.width_bytes) o += (int64_t)D.n_values * C.width_bytes;
            else if (C.phys_type == PQD_BOOLEAN) o += (int64_t)D.n_values;
            else o += (int64_t)D.chars + 4 * ((int64_t)D.n_values + 1);
        }
        *bytes_out = o;
    }
    return PQD_OK;
}

const void* pqd_plan_device_ptr(pqd_plan* pl, size_t c, pqd_output which) {
    if (!pl || c >= pl->chunks.size() || (int)which < 0 || (int)which >= N_OUT) return nullptr;
    return pl->outs[c].p[which];
}

int64_t pqd_plan_fetch(pqd_plan* pl, size_t c, pqd_output which, void* dst, size_t cap) {
    if (!pl || c >= pl->chunks.size() || (int)which < 0 || (int)which >= N_OUT || !pl->synced) return PQD_E_ARG;
    pqd_ctx* ctx = pl->ctx;
    const u8* p = pl->outs[c].p[which];
    if (!p) return 0;
    pqd_chunk_info ci;
    pqd_plan_chunk_info(pl, c, &ci);
    size_t n = 0;
    switch (which) {
    case PQD_OUT_DEFS: n = (size_t)ci.n_levels; break;
    case PQD_OUT_REPS: n = (size_t)ci.n_levels; break;
    case PQD_OUT_VALUES: n = (size_t)ci.value_bytes; break;
    case PQD_OUT_OFFSETS: n = ((size_t)ci.n_values + 1) * 4; break;
    case PQD_OUT_CHARS: n = (size_t)ci.chars_bytes; break;
    case PQD_OUT_VALIDITY: n = ((size_t)ci.n_levels + 7) / 8; break;
    case PQD_OUT_PAGE_NONNULL: n = (size_t)ci.n_pages * 8; break;
    case PQD_OUT_ROW_LEVEL_OFF: case PQD_OUT_ROW_VALUE_OFF: n = ((size_t)ci.n_rows + 1) * 4; break;
    case PQD_OUT_VALUE_INDEX: n = (((size_t)ci.n_levels + 63) / 64 + 1) * 4; break;
    }
    if (n > pl->outs[c].cap[which]) n = pl->outs[c].cap[which];
    if (n > cap) return fail(ctx, PQD_E_ARG, "destination too small");
    if (n == 0) return 0;
    if (!dst) return PQD_E_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemcpyAsync(dst, p, n, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return (int64_t)n;
}

int pqd_plan_scatter(pqd_plan* pl, size_t c, void* dst_device, size_t dst_bytes) {
    if (!pl || c >= pl->chunks.size() || !pl->ran) return PQD_E_ARG;
    pqd_ctx* ctx = pl->ctx;
    const ChunkRec& C = pl->chunks[c];
    if (C.max_def <= 0) return fail(ctx, PQD_E_ARG, "scatter: the column has no definition levels (its values are its slots)");
    const bool strings = C.phys_type == PQD_BYTE_ARRAY;
    const int w = strings ? 4 : C.phys_type == PQD_BOOLEAN ? 1 : (int)C.width_bytes;
    if (w != 1 && w != 4 && w != 8) return fail(ctx, PQD_E_UNSUPPORTED, "scatter: value width");
    const u64 n_slots = C.n_levels + (strings ? 1 : 0);
    if (!dst_device || ((uintptr_t)dst_device & 15) || dst_bytes < n_slots * (u64)w)
        return fail(ctx, PQD_E_ARG, "scatter: destination missing, not 16-byte aligned or too small");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t src_cap = pl->outs[c].cap[strings ? PQD_OUT_OFFSETS : PQD_OUT_VALUES] / (size_t)w;
    HIP_TRY(ctx, launch_scatter(ctx->stream, w, strings, C.validity, C.value_index, strings ? C.offsets : C.values, (u8*)dst_device,
                                n_slots, (C.n_levels + 63) / 64, (u32)(src_cap < 0xFFFFFFFFull ? src_cap : 0xFFFFFFFFull)));
    return PQD_OK;
}

int64_t pqd_plan_h2d_bytes(pqd_plan* pl) { return pl ? pl->h2d_bytes : PQD_E_ARG; }
int64_t pqd_plan_shared_reps(pqd_plan* pl) { return pl ? pl->n_shared_rep : PQD_E_ARG; }

int64_t pqd_plan_string_fallbacks(pqd_plan* pl) {
    if (!pl || !pl->synced) return PQD_E_ARG;
    int64_t n = 0;
    for (u32 f : pl->h_str_failed) n += f != 0;
    return n;
}

int64_t pqd_plan_debug_stamps(pqd_plan* pl, void* dst, size_t cap) {
    if (pl && !pl->lv.stamps && pl->schain.str_stamps) {  // PQD_STR_STAMPS without PQD_LV_STAMPS: the string kernel's stamps
        const size_t n = (size_t)pl->schain.n_units * SF_STAMPS * 8;
        if (n > cap || !dst) return PQD_E_ARG;
        HIP_TRY(pl->ctx, hipSetDevice(pl->ctx->device));
        HIP_TRY(pl->ctx, hipMemcpy(dst, pl->schain.str_stamps, n, hipMemcpyDeviceToHost));
        return (int64_t)n;
    }
    if (!pl || !pl->lv.stamps) return 0;
    const size_t n = pl->lv_order.size() * LV_STAMPS * 8;  // positions without a super-tile stay zero
    if (n > cap || !dst) return PQD_E_ARG;
    HIP_TRY(pl->ctx, hipSetDevice(pl->ctx->device));
    HIP_TRY(pl->ctx, hipMemcpy(dst, pl->lv.stamps, n, hipMemcpyDeviceToHost));
    return (int64_t)n;
}

int pqd_decode(pqd_ctx* ctx, const pqd_chunk_desc* chunks, size_t n_chunks, const pqd_page_desc* pages, size_t n_pages,
               const void* src_host, size_t src_len, pqd_plan** out) {
    if (!ctx || !out || !src_host || (!chunks && n_chunks) || (!pages && n_pages)) return PQD_E_ARG;
    // SNAPPY chunks: their pages are decoded on the device into an area behind the file bytes, and the
    // page decoder then reads them there like uncompressed pages (pageData, fields.go:358-402).
    std::vector<pqd_chunk_desc> cv(chunks, chunks + n_chunks);
    std::vector<pqd_page_desc> pv(pages, pages + n_pages);
    // Every page as the caller describes it must lie inside the host buffer.  Below, SNAPPY pages move to an area
    // behind the file bytes and the plan is made for that larger range, so this is the only place where the
    // descriptors of the uncompressed chunks are checked against the real length (the planner reads level block
    // lengths from src_host at these offsets).
    for (size_t c = 0; c < n_chunks; c++) {
        if (chunks[c].first_page < 0 || chunks[c].n_pages < 0 || (size_t)chunks[c].first_page + (size_t)chunks[c].n_pages > n_pages)
            return fail(ctx, PQD_E_ARG, "chunk page range out of bounds");
        for (int k = 0; k < chunks[c].n_pages; k++) {
            const pqd_page_desc& pd = pages[(size_t)chunks[c].first_page + k];
            if (pd.src_off < 0 || pd.payload_len < 0 || (uint64_t)pd.src_off + (uint64_t)pd.payload_len > src_len)
                return fail(ctx, PQD_E_CORRUPT, "page payload outside the source buffer");
        }
    }
    // Only the bytes that the described pages cover go to the device (readRowGroup consumes its own row group's bytes
    // and nothing else, template.go:320-351): [lo, hi) = from the first payload byte to the last, page headers between
    // them included.  Everything below works on offsets relative to lo.
    uint64_t lo = src_len, hi = 0;
    for (size_t c = 0; c < n_chunks; c++)
        for (int k = 0; k < chunks[c].n_pages; k++) {
            const pqd_page_desc& pd = pages[(size_t)chunks[c].first_page + k];
            lo = std::min<uint64_t>(lo, (uint64_t)pd.src_off);
            hi = std::max<uint64_t>(hi, (uint64_t)pd.src_off + (uint64_t)pd.payload_len);
        }
    if (hi < lo) lo = hi = 0;  // no pages
    const u8* const host = (const u8*)src_host + lo;
    const size_t span = (size_t)(hi - lo);
    for (pqd_page_desc& pd : pv) pd.src_off -= (int64_t)lo;
    std::vector<pqd_snappy_block> blocks;
    std::vector<size_t> block_page;
    for (size_t c = 0; c < n_chunks; c++) {
        if (cv[c].codec == 0) continue;
        if (cv[c].codec != 1) return fail(ctx, PQD_E_UNSUPPORTED, "codec other than UNCOMPRESSED and SNAPPY");
        for (int k = 0; k < cv[c].n_pages; k++) {
            const size_t pi = (size_t)cv[c].first_page + k;
            pqd_snappy_block b;
            memset(&b, 0, sizeof b);
            b.src_off = pv[pi].src_off;
            b.comp_len = pv[pi].payload_len;
            blocks.push_back(b);
            block_page.push_back(pi);
        }
        cv[c].codec = 0;
    }
    const size_t file_area = align_up(span + 64, 256);
    size_t total = file_area;
    if (!blocks.empty()) {
        int rc = pqd_snappy_sizes(host, span, blocks.data(), blocks.size());
        if (rc) return fail(ctx, rc, "bad snappy preamble or page outside the file");
        for (size_t i = 0; i < blocks.size(); i++) {
            if (blocks[i].decoded_len > 0x7fffffff) return fail(ctx, PQD_E_UNSUPPORTED, "snappy page of 2 GiB or more");
            blocks[i].dst_off = (int64_t)(total - file_area);
            pqd_page_desc& pd = pv[block_page[i]];
            pd.src_off = (int64_t)total;
            pd.payload_len = (int32_t)blocks[i].decoded_len;
            total += align_up((size_t)blocks[i].decoded_len + 16, 256);
        }
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    u8* dsrc = (u8*)pool_alloc(ctx, total + 64);
    if (!dsrc) return fail(ctx, PQD_E_NOMEM, "device allocation for the source failed");
    // queued, not waited for: the plan is made on the host while the bytes travel (from pageable memory the call
    // returns when the bytes have left the caller's buffer), and everything that reads them is queued behind the copy
    int rc = PQD_OK;
    if (span && hipMemcpyAsync(dsrc, host, span, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) rc = fail(ctx, PQD_E_HIP, "upload of the page bytes failed");
    if (!rc && !blocks.empty()) {
        rc = pqd_snappy_decode(ctx, blocks.data(), blocks.size(), dsrc, span, dsrc + file_area, total - file_area);
        // the level block sizes of decoded pages can only be read on the device
        if (!rc) rc = pqd_level_block_sizes(ctx, cv.data(), n_chunks, pv.data(), n_pages, dsrc, total);
        if (!rc)
            for (size_t c = 0; c < n_chunks; c++)
                if (chunks[c].codec == 0)  // pages of uncompressed chunks: sizes come from the host copy as before
                    for (int k = 0; k < cv[c].n_pages; k++) {
                        pv[(size_t)cv[c].first_page + k].rep_bytes = pages[(size_t)cv[c].first_page + k].rep_bytes;
                        pv[(size_t)cv[c].first_page + k].def_bytes = pages[(size_t)cv[c].first_page + k].def_bytes;
                    }
    }
    pqd_plan* pl = nullptr;
    // the planner reads level block lengths of uncompressed pages from the host bytes: the same range, the same offsets
    if (!rc) rc = pqd_plan_create(ctx, cv.data(), n_chunks, pv.data(), n_pages, host, total, &pl);
    if (rc) {
        (void)hipStreamSynchronize(ctx->stream);
        pool_release(ctx, dsrc);
        return rc;
    }
    pl->h2d_bytes = (int64_t)span;
    pl->owned_src = dsrc;
    if ((rc = pqd_plan_run(pl, pl->owned_src))) {
        pqd_plan_destroy(pl);
        return rc;
    }
    rc = pqd_plan_sync(pl);
    *out = pl;  // kept even on a decode error so the caller can inspect partial results
    return rc;
}

int pqd_rle_decode(pqd_ctx* ctx, int width, const void* block_, size_t block_len, int esz, void* out_host,
                   size_t max_values, int64_t* n_decoded) {
    if (!ctx || !block_ || !n_decoded || width < 0 || width > 32) return PQD_E_ARG;
    if (!(esz == 1 || esz == 2 || esz == 4) || width > 8 * esz) return fail(ctx, PQD_E_ARG, "element size too small for width");
    if (max_values > 0x7fffffffu) return fail(ctx, PQD_E_ARG, "max_values too large");
    const u8* block = (const u8*)block_;
    if (block_len < 4) return fail(ctx, PQD_E_CORRUPT, "level block shorter than its length field");
    int32_t len;
    memcpy(&len, block, 4);
    if (len < 0 || (size_t)len > block_len - 4) return fail(ctx, PQD_E_CORRUPT, "level block length outside the buffer");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    if (esz == 1) {
        // one byte per value: the single-pass level kernel of the page path, its tables in a scratch buffer
        ArenaPlan ap1;
        const size_t o_src = ap1.take((size_t)len + 64), o_out = ap1.take(max_values + 64);
        const size_t o_sc = ap1.take(lv_scratch_bytes((size_t)len)), o_res = ap1.take(16);
        u8* A1 = nullptr;
        hipError_t e1 = hipMalloc((void**)&A1, ap1.total);
        if (e1 != hipSuccess) return fail(ctx, PQD_E_NOMEM, "hipMalloc failed", e1);
        struct Free1 { u8* p; ~Free1() { (void)hipFree(p); } } fr1{A1};
        if (len) HIP_TRY(ctx, hipMemcpyAsync(A1 + o_src, block + 4, (size_t)len, hipMemcpyHostToDevice, st));
        HIP_TRY(ctx, launch_rle_decode_u8(st, A1 + o_sc, width, A1 + o_src, (u32)len, A1 + o_out, (u32)max_values, (long long*)(A1 + o_res)));
        long long res[2] = {0, 0};
        HIP_TRY(ctx, hipMemcpyAsync(res, A1 + o_res, 16, hipMemcpyDeviceToHost, st));
        HIP_TRY(ctx, hipStreamSynchronize(st));
        if (res[1]) return fail(ctx, (int)res[1], "malformed level block");
        *n_decoded = (int64_t)res[0];
        const u64 n_emit1 = (u64)res[0] < max_values ? (u64)res[0] : max_values;
        if (n_emit1 == 0) return PQD_OK;
        if (!out_host) return PQD_E_ARG;
        HIP_TRY(ctx, hipMemcpyAsync(out_host, A1 + o_out, n_emit1, hipMemcpyDeviceToHost, st));
        HIP_TRY(ctx, hipStreamSynchronize(st));
        return PQD_OK;
    }
    const u32 n_tiles = ((u32)len + CH_T - 1) / CH_T;
    const u32 n_otiles = (u32)((max_values + OT - 1) / OT);
    const u32 n_groups = (n_tiles + CH_G_LEVELS - 1) / CH_G_LEVELS;
    ArenaPlan ap;
    const size_t o_src = ap.take((size_t)len + 64);
    const size_t o_reg = ap.take(sizeof(Region));
    const size_t o_tp = ap.take(8), o_otp = ap.take(8), o_gp = ap.take(8);
    const size_t o_exit = ap.take((size_t)n_tiles * CH_E_LEVELS * sizeof(ExitRec));
    const size_t o_gexit = ap.take((size_t)n_groups * CH_E_LEVELS * sizeof(ExitRec));
    const size_t o_tot = ap.take(sizeof(RegionTotal));
    const size_t o_runs = ap.take((size_t)len * sizeof(RunRec));
    const size_t o_out = ap.take(max_values * (size_t)esz + 64);
    const size_t o_ff = ap.total;
    const size_t o_te = ap.take((size_t)n_tiles * sizeof(TileEntry));
    const size_t o_ge = ap.take((size_t)n_groups * sizeof(TileEntry));
    const size_t o_otf = ap.take(((size_t)n_otiles + 1) * 4);
    const size_t ff_end = ap.total;
    const size_t o_zero = ap.total;
    const size_t o_pdyn = ap.take(sizeof(PageDyn));
    const size_t o_status = ap.take(16);
    u8* A = nullptr;
    hipError_t e = hipMalloc((void**)&A, ap.total);
    if (e != hipSuccess) return fail(ctx, PQD_E_NOMEM, "hipMalloc failed", e);
    struct Free { u8* p; ~Free() { (void)hipFree(p); } } fr{A};
    Region R;
    memset(&R, 0, sizeof R);
    R.src_off = 0; R.len = (u32)len; R.width = (u32)width; R.n_out = 0; R.out = A + o_out; R.count_level = ERR_POS;
    const u32 tp[2] = {0, n_tiles}, otp[2] = {0, n_otiles}, gp[2] = {0, n_groups};
    const ChainArgs ca{(Region*)(A + o_reg), (u32*)(A + o_tp), (u32*)(A + o_gp), 1, n_tiles, n_groups, (ExitRec*)(A + o_exit),
                       (ExitRec*)(A + o_gexit), (TileEntry*)(A + o_te), (TileEntry*)(A + o_ge), (RegionTotal*)(A + o_tot)};
    HIP_TRY(ctx, hipMemsetAsync(A + o_ff, 0xFF, ff_end - o_ff, st));
    HIP_TRY(ctx, hipMemsetAsync(A + o_zero, 0, ap.total - o_zero, st));
    HIP_TRY(ctx, hipMemcpyAsync(A + o_src, block + 4, (size_t)len, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(A + o_reg, &R, sizeof R, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(A + o_tp, tp, 8, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(A + o_otp, otp, 8, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(A + o_gp, gp, 8, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, launch_chain(st, KIND_LEVELS, A + o_src, ca, (RunRec*)(A + o_runs), (u32*)(A + o_otf), (PageDyn*)(A + o_pdyn),
                              (int*)(A + o_status), 0));
    RegionTotal tot;
    int status[4];
    HIP_TRY(ctx, hipMemcpyAsync(&tot, A + o_tot, sizeof tot, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipMemcpyAsync(status, A + o_status, 16, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    if (status[0] || tot.err) return fail(ctx, PQD_E_CORRUPT, "malformed level block");
    *n_decoded = (int64_t)tot.vals;
    const u64 n_emit = tot.vals < max_values ? tot.vals : max_values;
    if (n_emit == 0) return PQD_OK;
    if (!out_host) return PQD_E_ARG;
    R.n_out = n_emit;
    const u32 otp2[2] = {0, (u32)((n_emit + OT - 1) / OT)};
    HIP_TRY(ctx, hipMemcpyAsync(A + o_reg, &R, sizeof R, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(A + o_otp, otp2, 8, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, launch_chain(st, KIND_LEVELS, A + o_src, ca, (RunRec*)(A + o_runs), (u32*)(A + o_otf), (PageDyn*)(A + o_pdyn),
                              (int*)(A + o_status), 1));
    HIP_TRY(ctx, launch_levels_expand(st, esz, A + o_src, (Region*)(A + o_reg), nullptr, (u32*)(A + o_otp), 1, otp2[1], (RunRec*)(A + o_runs),
                                      (u32*)(A + o_otf), (RegionTotal*)(A + o_tot), nullptr, (PageDyn*)(A + o_pdyn)));
    HIP_TRY(ctx, hipMemcpyAsync(out_host, A + o_out, n_emit * (size_t)esz, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    return PQD_OK;
}

int pqd_rle_decode_noalloc(pqd_ctx* ctx, int width, const void* runs_device, size_t runs_len, void* out_device,
                           size_t max_values, int64_t* result_device) {
    if (!ctx || !result_device || (!runs_device && runs_len) || (!out_device && max_values)) return PQD_E_ARG;
    if (width < 0 || width > 8) return fail(ctx, PQD_E_ARG, "width must be 0..8 (one byte per value)");
    if (runs_len > PQD_NOALLOC_MAX_BLOCK) return fail(ctx, PQD_E_ARG, "level block larger than PQD_NOALLOC_MAX_BLOCK (16 MiB)");
    if (max_values > 0x7fffffffu) return fail(ctx, PQD_E_ARG, "max_values too large");
    if (!lv_scratch_ptr(ctx->device)) return fail(ctx, PQD_E_HIP, "static scratch not initialised on this device");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, launch_rle_decode_u8_static(ctx->device, ctx->stream, width, (const u8*)runs_device, (u32)runs_len, (u8*)out_device,
                                             (u32)max_values, (long long*)result_device));
    return PQD_OK;
}

// ---------------------------------------------------------------------------------------------
// SNAPPY pages (pageData's SNAPPY arm, fields.go:361-371): decompressed on the device into a buffer the
// page decoder then reads like an uncompressed file.
int pqd_snappy_sizes(const void* src_host_, size_t src_len, pqd_snappy_block* blocks, size_t n) {
    if (!src_host_ || (!blocks && n)) return PQD_E_ARG;
    const u8* src = (const u8*)src_host_;
    for (size_t i = 0; i < n; i++) {
        pqd_snappy_block& b = blocks[i];
        if (b.src_off < 0 || b.comp_len < 0 || (u64)b.src_off + (u64)b.comp_len > src_len) return PQD_E_ARG;
        // decodedLen: binary.Uvarint, corrupt when it does not fit 32 bits
        u64 v = 0;
        unsigned shift = 0;
        int hdr = 0;
        for (int64_t k = 0; k < b.comp_len && k < 10; k++) {
            const u8 c = src[b.src_off + k];
            if (c < 0x80) {
                if (k == 9 && c > 1) break;
                v |= (u64)c << shift;
                hdr = (int)k + 1;
                break;
            }
            v |= (u64)(c & 0x7f) << shift;
            shift += 7;
        }
        if (hdr == 0 || v > 0xffffffffull) return PQD_E_CORRUPT;
        b.decoded_len = (int64_t)v;
        b.hdr_len = hdr;
    }
    return PQD_OK;
}

int pqd_snappy_decode(pqd_ctx* ctx, const pqd_snappy_block* blocks, size_t n, const void* src_device, size_t src_len,
                      void* dst_device, size_t dst_len) {
    if (!ctx || (!blocks && n) || !src_device || (!dst_device && dst_len)) return PQD_E_ARG;
    if (n == 0) return PQD_OK;
    if (n > 0x7fffffffu) return fail(ctx, PQD_E_ARG, "too many blocks");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    std::vector<Region> regions(n);
    std::vector<u32> tile_prefix(1, 0), group_prefix(1, 0), otile_region;
    u64 run_base = 0, n_otiles = 0;
    for (size_t i = 0; i < n; i++) {
        const pqd_snappy_block& b = blocks[i];
        if (b.src_off < 0 || b.comp_len < 0 || b.hdr_len < 1 || b.hdr_len > b.comp_len || b.decoded_len < 0 || b.dst_off < 0 ||
            (u64)b.src_off + (u64)b.comp_len > src_len || (u64)b.dst_off + (u64)b.decoded_len > dst_len)
            return fail(ctx, PQD_E_ARG, "snappy block outside its buffer");
        if (b.comp_len > 0x7fffffff || b.decoded_len > 0x7fffffff) return fail(ctx, PQD_E_UNSUPPORTED, "snappy block of 2 GiB or more");
        if ((((uintptr_t)dst_device + (u64)b.dst_off) & 15) != 0) return fail(ctx, PQD_E_ARG, "snappy destination must be 16-byte aligned");
        Region& R = regions[i];
        memset(&R, 0, sizeof R);
        R.src_off = (u64)b.src_off + (u64)b.hdr_len;
        R.len = (u32)(b.comp_len - b.hdr_len);
        R.tile_base = tile_prefix.back();
        R.group_base = group_prefix.back();
        R.otile_base = (u32)n_otiles;
        R.run_base = run_base;
        R.n_out = (u64)b.decoded_len;
        R.out = (u8*)dst_device + b.dst_off;
        R.page = (u32)i;
        R.count_level = ERR_POS;
        const u64 nt = ((u64)R.len + CH_T - 1) / CH_T, ng = ((u64)R.len + (u64)CH_G_SNAPPY * CH_T - 1) / ((u64)CH_G_SNAPPY * CH_T);
        const u64 no = ((u64)b.decoded_len + SN_T - 1) / SN_T;
        if ((u64)tile_prefix.back() + nt > 0x7fffffffull || n_otiles + no > 0x7fffffffull)
            return fail(ctx, PQD_E_UNSUPPORTED, "batch too large for 32-bit tile indices");
        tile_prefix.push_back(tile_prefix.back() + (u32)nt);
        group_prefix.push_back(group_prefix.back() + (u32)ng);
        otile_region.insert(otile_region.end(), (size_t)no, (u32)i);
        n_otiles += no;
        run_base += R.len;
    }
    otile_region.push_back(ERR_POS);
    const u32 n_tiles = tile_prefix.back(), n_groups = group_prefix.back();
    ArenaPlan ap;
    const size_t o_reg = ap.take(n * sizeof(Region));
    const size_t o_tp = ap.take((n + 1) * 4), o_gp = ap.take((n + 1) * 4), o_otr = ap.take(otile_region.size() * 4);
    const size_t o_exit = ap.take((size_t)n_tiles * CH_E_LEVELS * sizeof(ExitRec));
    const size_t o_gexit = ap.take((size_t)n_groups * CH_E_LEVELS * sizeof(ExitRec));
    const size_t o_tot = ap.take(n * sizeof(RegionTotal));
    const size_t o_runs = ap.take((size_t)run_base * sizeof(RunRec));
    const size_t o_ff = ap.total;
    const size_t o_te = ap.take((size_t)n_tiles * sizeof(TileEntry));
    const size_t o_ge = ap.take((size_t)n_groups * sizeof(TileEntry));
    const size_t o_otf = ap.take(((size_t)n_otiles + 1) * 4);
    const size_t ff_end = ap.total;
    const size_t o_status = ap.take(16);
    const size_t o_done = ap.take(((size_t)n_otiles + 1) * 4);  // a flag per output tile and the ticket counter
    const size_t zero_end = ap.total;
    u8* A = nullptr;
    hipError_t e = hipMalloc((void**)&A, ap.total);
    if (e != hipSuccess) return fail(ctx, PQD_E_NOMEM, "hipMalloc of the snappy scratch failed", e);
    struct Free { u8* p; ~Free() { (void)hipFree(p); } } fr{A};
    const ChainArgs ca{(Region*)(A + o_reg), (u32*)(A + o_tp), (u32*)(A + o_gp), (u32)n, n_tiles, n_groups, (ExitRec*)(A + o_exit),
                       (ExitRec*)(A + o_gexit), (TileEntry*)(A + o_te), (TileEntry*)(A + o_ge), (RegionTotal*)(A + o_tot)};
    HIP_TRY(ctx, hipMemsetAsync(A + o_ff, 0xFF, ff_end - o_ff, st));
    HIP_TRY(ctx, hipMemsetAsync(A + o_status, 0, zero_end - o_status, st));
    HIP_TRY(ctx, hipMemcpyAsync(A + o_reg, regions.data(), n * sizeof(Region), hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(A + o_tp, tile_prefix.data(), (n + 1) * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(A + o_gp, group_prefix.data(), (n + 1) * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(A + o_otr, otile_region.data(), otile_region.size() * 4, hipMemcpyHostToDevice, st));
    const u8* src = (const u8*)src_device;
    HIP_TRY(ctx, launch_chain(st, KIND_SNAPPY, src, ca, (RunRec*)(A + o_runs), (u32*)(A + o_otf), nullptr, (int*)(A + o_status), 0));
    HIP_TRY(ctx, launch_chain(st, KIND_SNAPPY, src, ca, (RunRec*)(A + o_runs), (u32*)(A + o_otf), nullptr, (int*)(A + o_status), 1));
    HIP_TRY(ctx, launch_snappy_tiles(st, src, (Region*)(A + o_reg), (u32*)(A + o_otr), (u32)n_otiles, (RunRec*)(A + o_runs),
                                     (u32*)(A + o_otf), (RegionTotal*)(A + o_tot), (int*)(A + o_status), (u32*)(A + o_done)));
    int status[4] = {0, 0, 0, 0};
    HIP_TRY(ctx, hipMemcpyAsync(status, A + o_status, 16, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    if (status[0]) {
        char b[160];
        snprintf(b, sizeof b, "snappy decode failed: device status %d at block %d", status[0], status[1]);
        ctx->err = b;
        return map_status(status[0]);
    }
    return PQD_OK;
}

int pqd_level_block_sizes(pqd_ctx* ctx, const pqd_chunk_desc* chunks, size_t n_chunks, pqd_page_desc* pages, size_t n_pages,
                          const void* src_device, size_t src_len) {
    if (!ctx || (!chunks && n_chunks) || (!pages && n_pages) || !src_device) return PQD_E_ARG;
    if (n_pages == 0) return PQD_OK;
    if (n_pages > 0x7fffffffu || n_chunks > 0x7fffffffu) return fail(ctx, PQD_E_ARG, "too many chunks or pages");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    std::vector<ChunkRec> C(n_chunks);
    std::vector<PageRec> P(n_pages);
    std::vector<char> seen(n_pages, 0);
    for (size_t c = 0; c < n_chunks; c++) {
        memset(&C[c], 0, sizeof(ChunkRec));
        C[c].max_def = chunks[c].max_def;
        C[c].max_rep = chunks[c].max_rep;
        if (chunks[c].first_page < 0 || chunks[c].n_pages < 0 || (size_t)chunks[c].first_page + (size_t)chunks[c].n_pages > n_pages)
            return fail(ctx, PQD_E_ARG, "chunk page range out of bounds");
        for (int k = 0; k < chunks[c].n_pages; k++) {
            const size_t pi = (size_t)chunks[c].first_page + k;
            const pqd_page_desc& pd = pages[pi];
            if (pd.src_off < 0 || pd.payload_len < 0 || (u64)pd.src_off + (u64)pd.payload_len > src_len)
                return fail(ctx, PQD_E_CORRUPT, "page payload outside the source buffer");
            memset(&P[pi], 0, sizeof(PageRec));
            P[pi].val_src_off = (u64)pd.src_off;
            P[pi].val_len = (u32)pd.payload_len;
            P[pi].chunk = (u32)c;
            seen[pi] = 1;
        }
    }
    for (size_t p = 0; p < n_pages; p++)
        if (!seen[p]) return fail(ctx, PQD_E_ARG, "page not owned by any chunk");
    ArenaPlan ap;
    const size_t o_c = ap.take(n_chunks * sizeof(ChunkRec)), o_p = ap.take(n_pages * sizeof(PageRec)), o_s = ap.take(n_pages * 8);
    u8* A = nullptr;
    hipError_t e = hipMalloc((void**)&A, ap.total);
    if (e != hipSuccess) return fail(ctx, PQD_E_NOMEM, "hipMalloc failed", e);
    struct Free { u8* p; ~Free() { (void)hipFree(p); } } fr{A};
    HIP_TRY(ctx, hipMemcpyAsync(A + o_c, C.data(), n_chunks * sizeof(ChunkRec), hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(A + o_p, P.data(), n_pages * sizeof(PageRec), hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, launch_level_block_sizes(st, (const u8*)src_device, (PageRec*)(A + o_p), (ChunkRec*)(A + o_c), (u32)n_pages,
                                          (int*)(A + o_s)));
    std::vector<int> sizes(n_pages * 2);
    HIP_TRY(ctx, hipMemcpyAsync(sizes.data(), A + o_s, n_pages * 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    for (size_t p = 0; p < n_pages; p++) {
        if (sizes[2 * p] < 0 || sizes[2 * p + 1] < 0) return fail(ctx, PQD_E_CORRUPT, "level block runs past the page payload");
        pages[p].rep_bytes = sizes[2 * p];
        pages[p].def_bytes = sizes[2 * p + 1];
    }
    return PQD_OK;
}

// ---------------------------------------------------------------------------------------------
// Level ENCODE, the mirror of pqd_rle_decode: rle.(*RLE).Write + Bytes (internal/rle/rle.go:48-156).
int pqd_rle_encode(pqd_ctx* ctx, int width, const void* levels_host, size_t n, void* out_host, size_t out_cap, int64_t* out_len) {
    if (!ctx || (!levels_host && n) || !out_len || width < 0 || width > 8) return PQD_E_ARG;
    if (n > 0x7ffffff0u) return fail(ctx, PQD_E_ARG, "too many levels");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const u32 n_tiles = (u32)((n + EN_T - 1) / EN_T);
    const u32 slots = (u32)(n / 8 + 2);
    // every element is at most 5 header bytes + 1 value byte or 1 header byte + `width` bytes
    const size_t out_bound = 4 + (size_t)slots * (size_t)(width > 5 ? width + 1 : 6);
    ArenaPlan ap;
    const size_t o_src = ap.take(n + 64);
    const size_t o_blk = ap.take(sizeof(EncBlock));
    const size_t o_tp = ap.take(8), o_ep = ap.take(8);
    const size_t o_lead = ap.take((size_t)n_tiles * 4), o_ext = ap.take(((size_t)n_tiles + 1) * 4);
    const size_t o_exits = ap.take((size_t)n_tiles * EN_E * 8);
    const size_t o_ne = ap.take(4), o_ol = ap.take(4);
    const u32 n_lchunks = (slots + EN_LAYOUT_CHUNK - 1) / EN_LAYOUT_CHUNK;
    const size_t o_el = ap.take((size_t)slots * 4), o_lp = ap.take(8), o_ch = ap.take((size_t)n_lchunks * 16);
    const size_t o_out = ap.take(out_bound + 64);
    const size_t o_ent = ap.take((size_t)n_tiles * 8);
    const size_t o_segs = ap.take((size_t)n_tiles * EN_SEG_RECS * 4);
    u8* A = nullptr;
    hipError_t e = hipMalloc((void**)&A, ap.total);
    if (e != hipSuccess) return fail(ctx, PQD_E_NOMEM, "hipMalloc failed", e);
    struct Free { u8* p; ~Free() { (void)hipFree(p); } } fr{A};
    EncBlock B;
    memset(&B, 0, sizeof B);
    B.src_off = 0; B.n = (u32)n; B.width = (u32)width; B.elem_base = 0; B.out = A + o_out;
    const u32 tp[2] = {0, n_tiles}, ep[2] = {0, slots}, lp[2] = {0, n_lchunks};
    HIP_TRY(ctx, hipMemsetAsync(A + o_ent, 0xFF, (size_t)n_tiles * 8 + 1, st));
    HIP_TRY(ctx, hipMemsetAsync(A + o_ext, 0, ((size_t)n_tiles + 1) * 4, st));
    if (n) HIP_TRY(ctx, hipMemcpyAsync(A + o_src, levels_host, n, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(A + o_blk, &B, sizeof B, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(A + o_tp, tp, 8, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(A + o_ep, ep, 8, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(A + o_lp, lp, 8, hipMemcpyHostToDevice, st));
    EncArgs a;
    a.blocks = (EncBlock*)(A + o_blk); a.tile_prefix = (u32*)(A + o_tp); a.elem_prefix = (u32*)(A + o_ep);
    a.n_blocks = 1; a.n_tiles = n_tiles; a.elem_slots = slots;
    a.lead = (u32*)(A + o_lead); a.ext = (u32*)(A + o_ext); a.exits = A + o_exits; a.entries = A + o_ent;
    a.segs = (u32*)(A + o_segs);
    a.n_elems = (u32*)(A + o_ne); a.out_len = (u32*)(A + o_ol);
    a.elems = (u32*)(A + o_el);
    a.lchunk_prefix = (u32*)(A + o_lp); a.n_lchunks = n_lchunks; a.chunks = A + o_ch;
    HIP_TRY(ctx, launch_rle_encode(st, A + o_src, a));
    u32 L = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&L, A + o_ol, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    *out_len = (int64_t)L + 4;
    if ((size_t)L + 4 > out_bound) return fail(ctx, PQD_E_CORRUPT, "encoder overran its bound");
    if ((size_t)L + 4 > out_cap) return fail(ctx, PQD_E_ARG, "destination too small");
    if (!out_host) return PQD_E_ARG;
    HIP_TRY(ctx, hipMemcpyAsync(out_host, A + o_out, (size_t)L + 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    return PQD_OK;
}

// Page payloads of the write path (RequiredField.DoWrite / OptionalField.DoWrite, fields.go:95-111, 232-271,
// UNCOMPRESSED): [repetition levels][definition levels][PLAIN values] per page, plus the statistics the
// generated stats types keep.  Host arrays in, host payloads out; everything in between runs on the device.
int pqd_encode_pages(pqd_ctx* ctx, const pqd_enc_page* pages, size_t n, const void* src_host, size_t src_len, void* out_host,
                     size_t out_cap, pqd_enc_result* results, int64_t* out_len) {
    if (!ctx || (!pages && n) || (!src_host && src_len) || !results || !out_len) return PQD_E_ARG;
    *out_len = 0;
    if (n == 0) return PQD_OK;
    if (n > 0x3fffffffu) return fail(ctx, PQD_E_ARG, "too many pages");
    const u8* hs = (const u8*)src_host;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    std::vector<EncPage> P(n);
    std::vector<EncBlock> B;
    std::vector<u32> tile_prefix(1, 0), elem_prefix(1, 0), lchunk_prefix(1, 0), chunk_prefix(1, 0);
    size_t bound = 0;
    auto add_block = [&](int64_t off, int64_t nl, int max_level) -> int {
        if (off < 0 || (off & 3) || (u64)off + (u64)nl > src_len) return -1;
        EncBlock b;
        memset(&b, 0, sizeof b);
        b.src_off = (u64)off; b.n = (u32)nl; b.width = (u32)bits_len((unsigned)max_level);
        b.elem_base = elem_prefix.back();
        const u64 slots = (u64)nl / 8 + 2, tiles = ((u64)nl + EN_T - 1) / EN_T;
        if ((u64)elem_prefix.back() + slots > 0x7fffffffull || (u64)tile_prefix.back() + tiles > 0x7fffffffull) return -2;
        elem_prefix.push_back(elem_prefix.back() + (u32)slots);
        lchunk_prefix.push_back(lchunk_prefix.back() + (u32)((slots + EN_LAYOUT_CHUNK - 1) / EN_LAYOUT_CHUNK));
        tile_prefix.push_back(tile_prefix.back() + (u32)tiles);
        bound += 4 + (size_t)slots * (b.width > 5 ? b.width + 1 : 6);
        B.push_back(b);
        return (int)B.size() - 1;
    };
    for (size_t i = 0; i < n; i++) {
        const pqd_enc_page& q = pages[i];
        EncPage& e = P[i];
        memset(&e, 0, sizeof e);
        int w = 0;
        switch (q.phys_type) {
        case PQD_INT32: case PQD_FLOAT: w = 4; break;
        case PQD_INT64: case PQD_DOUBLE: w = 8; break;
        case PQD_BOOLEAN: case PQD_BYTE_ARRAY: break;
        default: return fail(ctx, PQD_E_UNSUPPORTED, "unsupported physical type");
        }
        if (q.max_def < 0 || q.max_def > 255 || q.max_rep < 0 || q.max_rep > q.max_def || q.n_levels < 0 || q.n_values < 0 ||
            q.n_levels > 0x7ffffff0 || q.n_values > 0x7ffffff0 || (q.max_def > 0 ? q.n_values > q.n_levels : q.n_values != q.n_levels))
            return fail(ctx, PQD_E_ARG, "bad page description");
        e.phys_type = q.phys_type; e.width = (u32)w; e.n_values = (u64)q.n_values;
        e.stats_kind = (q.phys_type == PQD_FLOAT || q.phys_type == PQD_DOUBLE) ? 2u : q.is_unsigned ? 1u : 0u;
        e.rep_block = e.def_block = 0xFFFFFFFFu;
        if (q.max_rep > 0) {
            const int r = add_block(q.reps_off, q.n_levels, q.max_rep);
            if (r < 0) return fail(ctx, r == -1 ? PQD_E_ARG : PQD_E_UNSUPPORTED, "repetition levels: bad offset (4-byte aligned, inside the source) or batch too large");
            e.rep_block = (u32)r;
        }
        if (q.max_def > 0) {
            const int r = add_block(q.defs_off, q.n_levels, q.max_def);
            if (r < 0) return fail(ctx, r == -1 ? PQD_E_ARG : PQD_E_UNSUPPORTED, "definition levels: bad offset (4-byte aligned, inside the source) or batch too large");
            e.def_block = (u32)r;
        }
        u64 vb, units;
        if (w) {
            vb = (u64)q.n_values * (u64)w;
            if (q.values_off < 0 || (q.values_off & (w - 1)) || (u64)q.values_off + vb > src_len) return fail(ctx, PQD_E_ARG, "values outside the source or misaligned");
            units = (vb + ENC_VALUE_CHUNK - 1) / ENC_VALUE_CHUNK;
        } else if (q.phys_type == PQD_BOOLEAN) {
            vb = ((u64)q.n_values + 7) / 8;
            if (q.values_off < 0 || (u64)q.values_off + (u64)q.n_values > src_len) return fail(ctx, PQD_E_ARG, "values outside the source");
            units = (vb + ENC_VALUE_CHUNK - 1) / ENC_VALUE_CHUNK;
        } else {
            if (q.offsets_off < 0 || (q.offsets_off & 3) || (u64)q.offsets_off + 4 * ((u64)q.n_values + 1) > src_len)
                return fail(ctx, PQD_E_ARG, "string offsets outside the source or misaligned");
            const int32_t* offs = (const int32_t*)(hs + q.offsets_off);
            for (int64_t k = 0; k < q.n_values; k++)
                if (offs[k + 1] < offs[k]) return fail(ctx, PQD_E_ARG, "string offsets must not decrease");
            if (offs[0] < 0 || q.values_off < 0 || (u64)q.values_off + (u64)offs[q.n_values] > src_len)
                return fail(ctx, PQD_E_ARG, "string bytes outside the source");
            vb = 4 * (u64)q.n_values + (u64)(offs[q.n_values] - offs[0]);
            units = ((u64)q.n_values + ENC_VALUE_CHUNK / 16 - 1) / (ENC_VALUE_CHUNK / 16);
        }
        e.values_off = (u64)q.values_off; e.offsets_off = (u64)(q.offsets_off < 0 ? 0 : q.offsets_off);
        if ((u64)chunk_prefix.back() + units > 0x7fffffffull) return fail(ctx, PQD_E_UNSUPPORTED, "batch too large");
        chunk_prefix.push_back(chunk_prefix.back() + (u32)units);
        bound += (size_t)vb;
    }
    const u32 nb = (u32)B.size(), n_tiles = tile_prefix.back(), slots = elem_prefix.back();
    ArenaPlan ap;
    const size_t o_src = ap.take(src_len + 64);
    const size_t o_pg = ap.take(n * sizeof(EncPage)), o_blk = ap.take((size_t)nb * sizeof(EncBlock));
    const size_t o_tp = ap.take(((size_t)nb + 1) * 4), o_ep = ap.take(((size_t)nb + 1) * 4), o_cp = ap.take((n + 1) * 4);
    const size_t o_lead = ap.take((size_t)n_tiles * 4), o_ext = ap.take(((size_t)n_tiles + 1) * 4);
    const size_t o_exits = ap.take((size_t)n_tiles * EN_E * 8), o_ent = ap.take((size_t)n_tiles * 8);
    const size_t o_segs = ap.take((size_t)n_tiles * EN_SEG_RECS * 4);
    const size_t o_ne = ap.take((size_t)nb * 4), o_ol = ap.take((size_t)nb * 4);
    const size_t o_el = ap.take((size_t)slots * 4);
    const size_t o_lp = ap.take(((size_t)nb + 1) * 4), o_ch = ap.take((size_t)lchunk_prefix.back() * 16);
    const size_t o_out = ap.take(bound + 64);
    const size_t o_sp = ap.take((size_t)chunk_prefix.back() * 16);
    u8* A = nullptr;
    hipError_t e = hipMalloc((void**)&A, ap.total);
    if (e != hipSuccess) return fail(ctx, PQD_E_NOMEM, "hipMalloc of the encode arena failed", e);
    struct Free { u8* p; ~Free() { (void)hipFree(p); } } fr{A};
    HIP_TRY(ctx, hipMemsetAsync(A + o_ent, 0xFF, (size_t)n_tiles * 8 + 1, st));
    HIP_TRY(ctx, hipMemsetAsync(A + o_ext, 0, ((size_t)n_tiles + 1) * 4, st));
    if (src_len) HIP_TRY(ctx, hipMemcpyAsync(A + o_src, src_host, src_len, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(A + o_pg, P.data(), n * sizeof(EncPage), hipMemcpyHostToDevice, st));
    if (nb) HIP_TRY(ctx, hipMemcpyAsync(A + o_blk, B.data(), (size_t)nb * sizeof(EncBlock), hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(A + o_tp, tile_prefix.data(), ((size_t)nb + 1) * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(A + o_ep, elem_prefix.data(), ((size_t)nb + 1) * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(A + o_lp, lchunk_prefix.data(), ((size_t)nb + 1) * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(A + o_cp, chunk_prefix.data(), (n + 1) * 4, hipMemcpyHostToDevice, st));
    EncArgs a;
    a.blocks = (EncBlock*)(A + o_blk); a.tile_prefix = (u32*)(A + o_tp); a.elem_prefix = (u32*)(A + o_ep);
    a.n_blocks = nb; a.n_tiles = n_tiles; a.elem_slots = slots;
    a.lead = (u32*)(A + o_lead); a.ext = (u32*)(A + o_ext); a.exits = A + o_exits; a.entries = A + o_ent;
    a.segs = (u32*)(A + o_segs);
    a.n_elems = (u32*)(A + o_ne); a.out_len = (u32*)(A + o_ol);
    a.elems = (u32*)(A + o_el);
    a.lchunk_prefix = (u32*)(A + o_lp); a.n_lchunks = lchunk_prefix.back(); a.chunks = A + o_ch;
    HIP_TRY(ctx, launch_rle_encode_layout(st, A + o_src, a));
    HIP_TRY(ctx, launch_encode_pages(st, A + o_src, (EncPage*)(A + o_pg), (u32*)(A + o_cp), (u32)n, chunk_prefix.back(),
                                     (EncBlock*)(A + o_blk), (u32*)(A + o_ol), A + o_out, (u64*)(A + o_sp)));
    HIP_TRY(ctx, launch_rle_encode_write(st, A + o_src, a));
    HIP_TRY(ctx, hipMemcpyAsync(P.data(), A + o_pg, n * sizeof(EncPage), hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    const u64 total = P[n - 1].payload_off + P[n - 1].rep_bytes + P[n - 1].def_bytes + P[n - 1].val_bytes;
    *out_len = (int64_t)total;
    if (total > bound) return fail(ctx, PQD_E_CORRUPT, "encoder overran its bound");
    for (size_t i = 0; i < n; i++) {
        pqd_enc_result& r = results[i];
        memset(&r, 0, sizeof r);
        r.payload_off = (int64_t)P[i].payload_off;
        r.payload_len = (int64_t)(P[i].rep_bytes + P[i].def_bytes + P[i].val_bytes);
        r.rep_bytes = (int32_t)P[i].rep_bytes;
        r.def_bytes = (int32_t)P[i].def_bytes;
        r.null_count = pages[i].n_levels - pages[i].n_values;
        if (P[i].width && !(pages[i].max_def > 0 && pages[i].n_values == 0)) {  // optional stats: nil while nonNils == 0
            r.minmax_len = (int32_t)P[i].width;
            memcpy(r.min, &P[i].min_bits, 8);
            memcpy(r.max, &P[i].max_bits, 8);
        }
    }
    if (total > out_cap) return fail(ctx, PQD_E_ARG, "destination too small");
    if (total) {
        if (!out_host) return PQD_E_ARG;
        HIP_TRY(ctx, hipMemcpyAsync(out_host, A + o_out, total, hipMemcpyDeviceToHost, st));
        HIP_TRY(ctx, hipStreamSynchronize(st));
    }
    return PQD_OK;
}

}  // extern "C"
