// Multi-GPU sharding below the C ABI (see include/pqdecode.h).  The reference reader is one goroutine on one
// host; what shards naturally is the row group: readRowGroup starts from fresh Field objects
// (cmd/parquetgen/gen/template.go:329), so row groups are independent and whole row groups go to GPUs in
// contiguous ranges balanced by payload bytes.  Nothing of the data path is exchanged; the only exchange is the
// row count of every shard, from which each shard's global row offset follows:
//   one process, several GPUs   pqd_multi_*: one context and one host thread per device, an exclusive scan in
//                               the process;
//   one process per GPU         pqd_rccl_*: ncclAllGather of one int64 per rank over RCCL (xGMI), on the
//                               context's stream.  librccl is loaded on first use, so the decoder itself does not
//                               depend on it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/pqdecode.h"

extern "C" {

int pqd_shard_assign(const int64_t* weights, int32_t n, int32_t world, int32_t* first, int32_t* count) {
    if (n < 0 || world <= 0 || (!weights && n) || !first || !count) return PQD_E_ARG;
    // prefix sums of the weights; cut r goes to the boundary nearest to total * r / world, never before the cut
    // in front of it (the same rule as parquet_amd/shard.py, which the tests compare with)
    std::vector<double> prefix((size_t)n + 1, 0.0);
    for (int32_t i = 0; i < n; i++) {
        if (weights[i] < 0) return PQD_E_ARG;
        prefix[(size_t)i + 1] = prefix[i] + (double)weights[i];
    }
    const double total = prefix[n];
    std::vector<int32_t> cuts(1, 0);
    for (int32_t r = 1; r < world; r++) {
        const double target = total * r / world;
        int32_t i = 0;  // first index with prefix[i] >= target (numpy.searchsorted, side left)
        {
            int32_t lo = 0, hi = n + 1;
            while (lo < hi) {
                const int32_t mid = (lo + hi) / 2;
                if (prefix[mid] < target) lo = mid + 1; else hi = mid;
            }
            i = lo;
        }
        const int32_t ic = i < n ? i : n;
        auto dist = [&](int32_t k) { const double d = prefix[k] - target; return d < 0 ? -d : d; };
        if (i > 0 && dist(i - 1) <= dist(ic)) i -= 1;
        int32_t c = i > cuts.back() ? i : cuts.back();
        if (c > n) c = n;
        cuts.push_back(c);
    }
    cuts.push_back(n);
    for (int32_t r = 0; r < world; r++) {
        first[r] = cuts[r];
        count[r] = cuts[r + 1] - cuts[r];
    }
    return PQD_OK;
}

int pqd_file_shard(const pqd_file* f, int32_t world, int32_t* first, int32_t* count) {
    if (!f) return PQD_E_ARG;
    const int32_t n = pqd_file_num_row_groups(f), nc = pqd_file_num_columns(f);
    std::vector<int64_t> w((size_t)(n > 0 ? n : 0), 0);
    for (int32_t g = 0; g < n; g++)
        for (int32_t c = 0; c < nc; c++) {
            int64_t m[6];
            if (pqd_file_chunk_meta(f, g, c, m) != PQD_OK) break;  // a row group may hold fewer chunks
            w[g] += m[2];                                          // total_compressed_size: header + payload bytes
        }
    return pqd_shard_assign(w.data(), n, world, first, count);
}

}  // extern "C"

struct pqd_multi {
    std::vector<pqd_ctx*> ctx;
    std::vector<pqd_plan*> plans;
    std::vector<int32_t> first, count;
    std::vector<int64_t> rows, row_offset;
    std::vector<int> status;
    std::string err;
};

extern "C" {

int pqd_multi_create(const int* devices, int n, pqd_multi** out) {
    if (!devices || n <= 0 || !out) return PQD_E_ARG;
    pqd_multi* m = new pqd_multi();
    for (int i = 0; i < n; i++) {
        pqd_ctx* c = nullptr;
        const int rc = pqd_create(devices[i], &c);
        if (rc != PQD_OK) {
            for (pqd_ctx* x : m->ctx) pqd_destroy(x);
            delete m;
            return rc;
        }
        m->ctx.push_back(c);
    }
    m->plans.assign((size_t)n, nullptr);
    m->first.assign((size_t)n, 0); m->count.assign((size_t)n, 0);
    m->rows.assign((size_t)n, 0); m->row_offset.assign((size_t)n + 1, 0);
    m->status.assign((size_t)n, 0);
    *out = m;
    return PQD_OK;
}

static void multi_release_plans(pqd_multi* m) {
    for (pqd_plan*& p : m->plans) {
        if (p) pqd_plan_destroy(p);
        p = nullptr;
    }
}

void pqd_multi_destroy(pqd_multi* m) {
    if (!m) return;
    multi_release_plans(m);
    for (pqd_ctx* c : m->ctx) pqd_destroy(c);
    delete m;
}

int pqd_multi_decode_file(pqd_multi* m, const void* buf, size_t len) {
    if (!m || !buf) return PQD_E_ARG;
    multi_release_plans(m);
    const int n = (int)m->ctx.size();
    // every shard walks the file itself: the descriptor arrays belong to the file object
    std::vector<pqd_file*> files((size_t)n, nullptr);
    char ebuf[256] = {0};
    for (int i = 0; i < n; i++) {
        const int rc = pqd_file_open(buf, len, &files[i], ebuf, sizeof ebuf);
        if (rc != PQD_OK) {
            for (pqd_file* f : files) if (f) pqd_file_close(f);
            m->err = ebuf;
            return rc;
        }
    }
    int rc = pqd_file_shard(files[0], n, m->first.data(), m->count.data());
    if (rc == PQD_OK) {
        std::vector<std::thread> th;
        for (int i = 0; i < n; i++) {
            m->status[i] = PQD_OK;
            m->rows[i] = 0;
            for (int32_t g = m->first[i]; g < m->first[i] + m->count[i]; g++) m->rows[i] += pqd_file_row_group_rows(files[i], g);
            if (m->count[i] == 0) continue;
            th.emplace_back([m, i, &files, buf, len]() {
                const pqd_chunk_desc* ch = nullptr; const pqd_page_desc* pg = nullptr;
                size_t nc = 0, np = 0;
                int r = pqd_file_descriptors(files[i], m->first[i], m->count[i], &ch, &nc, &pg, &np);
                if (r == PQD_OK) r = pqd_decode(m->ctx[i], ch, nc, pg, np, buf, len, &m->plans[i]);
                m->status[i] = r;
            });
        }
        for (std::thread& t : th) t.join();
        // the exchange: row counts of all shards -> global row offset of each
        m->row_offset[0] = 0;
        for (int i = 0; i < n; i++) m->row_offset[i + 1] = m->row_offset[i] + m->rows[i];
        for (int i = 0; i < n && rc == PQD_OK; i++)
            if (m->status[i] != PQD_OK) {
                rc = m->status[i];
                m->err = std::string("shard ") + std::to_string(i) + ": " + pqd_last_error(m->ctx[i]);
            }
    }
    for (pqd_file* f : files) pqd_file_close(f);
    return rc;
}

int pqd_multi_shards(const pqd_multi* m) { return m ? (int)m->ctx.size() : 0; }
const char* pqd_multi_last_error(const pqd_multi* m) { return m ? m->err.c_str() : "no object"; }

int pqd_multi_shard(const pqd_multi* m, int shard, int32_t* rg_first, int32_t* rg_count, int64_t* rows, int64_t* row_offset,
                    pqd_plan** plan) {
    if (!m || shard < 0 || (size_t)shard >= m->ctx.size()) return PQD_E_ARG;
    if (rg_first) *rg_first = m->first[shard];
    if (rg_count) *rg_count = m->count[shard];
    if (rows) *rows = m->rows[shard];
    if (row_offset) *row_offset = m->row_offset[shard];
    if (plan) *plan = m->plans[shard];
    return PQD_OK;
}

// ---- one process per GPU: the row counts go through RCCL ------------------------------------------------------
struct Id128 { char b[128]; };  // ncclUniqueId, passed by value
namespace {
struct Rccl {
    void* lib = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, Id128, int) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
Rccl g_rccl;
std::once_flag g_rccl_once;
bool rccl_load() {
    std::call_once(g_rccl_once, []() {
        void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
        if (!h) return;
        g_rccl.GetUniqueId = (int (*)(void*))dlsym(h, "ncclGetUniqueId");
        g_rccl.CommInitRank = (int (*)(void**, int, Id128, int))dlsym(h, "ncclCommInitRank");
        g_rccl.CommDestroy = (int (*)(void*))dlsym(h, "ncclCommDestroy");
        g_rccl.AllGather = (int (*)(const void*, void*, size_t, int, void*, hipStream_t))dlsym(h, "ncclAllGather");
        g_rccl.GetErrorString = (const char* (*)(int))dlsym(h, "ncclGetErrorString");
        if (g_rccl.GetUniqueId && g_rccl.CommInitRank && g_rccl.CommDestroy && g_rccl.AllGather) g_rccl.lib = h;
    });
    return g_rccl.lib != nullptr;
}
}  // namespace

int pqd_rccl_unique_id(void* id128) {
    if (!id128) return PQD_E_ARG;
    if (!rccl_load()) return PQD_E_UNSUPPORTED;
    return g_rccl.GetUniqueId(id128) == 0 ? PQD_OK : PQD_E_HIP;
}

// A communicator owns what a gather needs, so that the gather itself allocates and creates nothing: the RCCL
// communicator, a stream, 8 (world + 1) bytes of device memory ([0] = this rank's count, [1 .. world] = everyone's)
// and the same in pinned host memory.
struct pqd_rccl_comm {
    void* nccl = nullptr;
    int device = 0, rank = 0, world = 1;
    hipStream_t stream = nullptr;
    int64_t* d = nullptr;
    int64_t* h = nullptr;
};

void pqd_rccl_comm_destroy(void* comm_) {
    pqd_rccl_comm* c = (pqd_rccl_comm*)comm_;
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) { (void)hipStreamSynchronize(c->stream); }
    if (c->nccl && rccl_load()) (void)g_rccl.CommDestroy(c->nccl);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    if (c->d) (void)hipFree(c->d);
    if (c->h) (void)hipHostFree(c->h);
    delete c;
}

int pqd_rccl_comm_create(int device, int rank, int world, const void* id128, void** comm) {
    if (!id128 || !comm || rank < 0 || world <= 0 || rank >= world) return PQD_E_ARG;
    if (!rccl_load()) return PQD_E_UNSUPPORTED;
    if (hipSetDevice(device) != hipSuccess) return PQD_E_HIP;
    pqd_rccl_comm* c = new pqd_rccl_comm();
    c->device = device; c->rank = rank; c->world = world;
    const size_t bytes = ((size_t)world + 1) * 8;
    Id128 id;
    memcpy(id.b, id128, sizeof id.b);
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess || hipMalloc((void**)&c->d, bytes) != hipSuccess ||
        hipHostMalloc((void**)&c->h, bytes, hipHostMallocDefault) != hipSuccess || g_rccl.CommInitRank(&c->nccl, world, id, rank) != 0) {
        pqd_rccl_comm_destroy(c);
        return PQD_E_HIP;
    }
    *comm = c;
    return PQD_OK;
}

int pqd_rccl_gather_rows(void* comm_, int64_t local_rows, int64_t* all_rows, int64_t* row_offset) {
    pqd_rccl_comm* c = (pqd_rccl_comm*)comm_;
    if (!c || !all_rows) return PQD_E_ARG;
    if (hipSetDevice(c->device) != hipSuccess) return PQD_E_HIP;
    c->h[0] = local_rows;
    if (hipMemcpyAsync(c->d, c->h, 8, hipMemcpyHostToDevice, c->stream) != hipSuccess) return PQD_E_HIP;
    if (g_rccl.AllGather(c->d, c->d + 1, 1, /* ncclInt64 */ 4, c->nccl, c->stream) != 0) return PQD_E_HIP;
    if (hipMemcpyAsync(c->h + 1, c->d + 1, (size_t)c->world * 8, hipMemcpyDeviceToHost, c->stream) != hipSuccess) return PQD_E_HIP;
    if (hipStreamSynchronize(c->stream) != hipSuccess) return PQD_E_HIP;
    int64_t o = 0;
    for (int r = 0; r < c->world; r++) {
        all_rows[r] = c->h[1 + r];
        if (r < c->rank) o += c->h[1 + r];
    }
    if (row_offset) *row_offset = o;
    return PQD_OK;
}

}  // extern "C"
