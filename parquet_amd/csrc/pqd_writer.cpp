// libpqsynth.so — writes Parquet files with the exact byte layout of the parsyl/parquet writer
// (uncompressed DATA_PAGE v1, PLAIN values, hybrid RLE levels).  Host-only support code: the decoder
// tests and the benchmark use it to synthesise reference-format inputs; it is not on the decode path.
//
// Layout followed (paths inside parsyl/parquet):
//   file      "PAR1", row groups, thrift FileMetaData, uint32 LE length, "PAR1"   parquet.go:217-260,
//                                                                               cmd/parquetgen/gen/template.go:128-133,189-196
//   page      thrift PageHeader{type=DATA_PAGE, sizes, DataPageHeader{num_values, PLAIN, RLE, RLE, stats}}
//             then [rep block][def block][PLAIN values of non-null entries]      parquet.go:154-186, fields.go:232-271
//   levels    int32 LE byte length + hybrid runs, width = bits.Len(max level)    fields.go:240,247,438-445
//   encoder   an RLE run is emitted for >= 8 equal values that start on a group boundary, everything
//             else goes into bit-packed groups of 8 (<= 63 groups per run), tail zero padded
//                                                                               internal/rle/rle.go:48-156
//   values    4/8-byte LE; bools bit i%8 of byte i/8, padded per page; strings uint32 LE length + bytes
//                                                                               template_required.go:41-46, template_bool.go:24-33,
//                                                                               template_string.go:29-36
//   schema    root{name "root"} then depth-first groups and leaves              parquet.go:41-85
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "thrift_compact.h"

namespace th = pqd::thrift;

namespace {

int bits_len(unsigned v) { int n = 0; while (v) { n++; v >>= 1; } return n; }

// Hybrid encoder working on whole groups: at a group boundary, eight equal values open an RLE run
// that absorbs every following equal value; otherwise the eight values (zero padded at the end of
// the input) are bit-packed.  This is the behaviour of the reference's per-value state machine.
class LevelEncoder {
  public:
    explicit LevelEncoder(int width) : w_(width) {}
    void encode(const uint8_t* v, size_t n, std::vector<uint8_t>& out) {
        size_t len_pos = out.size();
        out.insert(out.end(), 4, 0);
        size_t body = out.size();
        long hdr = -1;
        int groups = 0;
        size_t i = 0;
        auto close_run = [&]() {
            if (hdr >= 0) { out[(size_t)hdr] = (uint8_t)((groups << 1) | 1); hdr = -1; groups = 0; }
        };
        while (i < n) {
            bool rle = false;
            if (i + 8 <= n) {
                uint64_t x;
                memcpy(&x, v + i, 8);
                rle = x == (x & 0xFF) * 0x0101010101010101ull;
            }
            if (rle) {
                size_t j = i + 8;
                while (j < n && v[j] == v[i]) j++;
                close_run();
                uint64_t h = (uint64_t)(j - i) << 1;
                while (h >= 0x80) { out.push_back((uint8_t)(h | 0x80)); h >>= 7; }
                out.push_back((uint8_t)h);
                if (w_ > 0) out.push_back(v[i]);
                i = j;
            } else {
                if (groups >= 63) close_run();
                if (hdr < 0) { hdr = (long)out.size(); out.push_back(0); }
                uint8_t g[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                size_t m = std::min<size_t>(8, n - i);
                memcpy(g, v + i, m);
                if (w_ >= 1 && w_ <= 4) {
                    uint32_t acc = 0;
                    const uint32_t mask = (1u << w_) - 1;
                    for (int k = 0; k < 8; k++) acc |= (uint32_t)(g[k] & mask) << (k * w_);
                    for (int k = 0; k < w_; k++) out.push_back((uint8_t)(acc >> (8 * k)));
                }
                groups++;
                i += m;
            }
        }
        close_run();
        uint32_t L = (uint32_t)(out.size() - body);
        memcpy(&out[len_pos], &L, 4);
    }

  private:
    int w_;
};

struct Col {
    std::string path;
    std::vector<std::string> parts;
    std::vector<int> reps;  // repetition type per path element
    int phys = 0, conv = -1, max_def = 0, max_rep = 0;
};

struct ChunkAcc {
    int64_t num_values = 0, total_size = 0;
    bool used = false;
};

}  // namespace

struct pqw {
    std::vector<uint8_t> buf;
    std::vector<Col> cols;
    int stats_mode = 1;
    int codec = 0;  // 0 UNCOMPRESSED, 1 SNAPPY (every page of the file)
    bool in_rg = false;
    std::vector<ChunkAcc> cur;
    struct RG { int64_t rows; std::vector<ChunkAcc> chunks; };
    std::vector<RG> rgs;
    int64_t total_rows = 0;
    int last_col = -1;
    bool finished = false;
};

namespace {

// A small greedy snappy compressor (raw block format): 64 KiB blocks compressed independently, 4-byte
// hash matches, copy elements with 1- and 2-byte offsets.  It exists so that the synthetic files can carry
// SNAPPY pages like the files the reference writes by default (fields.go:69,180); any snappy decoder
// reads its output.
void snappy_literal(const uint8_t* p, size_t n, std::vector<uint8_t>& out) {
    if (n == 0) return;
    const size_t x = n - 1;
    if (x < 60) out.push_back((uint8_t)(x << 2));
    else if (x < 256) { out.push_back(60 << 2); out.push_back((uint8_t)x); }
    else if (x < 65536) { out.push_back(61 << 2); out.push_back((uint8_t)x); out.push_back((uint8_t)(x >> 8)); }
    else { out.push_back(62 << 2); out.push_back((uint8_t)x); out.push_back((uint8_t)(x >> 8)); out.push_back((uint8_t)(x >> 16)); }
    out.insert(out.end(), p, p + n);
}
void snappy_compress(const uint8_t* in, size_t n, std::vector<uint8_t>& out) {
    for (size_t v = n;;) {  // uvarint preamble
        if (v < 0x80) { out.push_back((uint8_t)v); break; }
        out.push_back((uint8_t)(v | 0x80));
        v >>= 7;
    }
    std::vector<int32_t> table(1 << 14);
    for (size_t b0 = 0; b0 < n; b0 += 65536) {
        const uint8_t* p = in + b0;
        const size_t len = std::min<size_t>(65536, n - b0);
        std::fill(table.begin(), table.end(), -1);
        size_t i = 0, lit = 0;
        while (i + 4 <= len) {
            uint32_t x;
            memcpy(&x, p + i, 4);
            const uint32_t h = (x * 0x1e35a7bdu) >> 18;
            const int32_t cand = table[h];
            table[h] = (int32_t)i;
            uint32_t y = 0;
            if (cand >= 0) memcpy(&y, p + cand, 4);
            if (cand < 0 || x != y) { i++; continue; }
            snappy_literal(p + lit, i - lit, out);
            size_t ml = 4;
            while (i + ml < len && p[cand + ml] == p[i + ml]) ml++;
            const size_t off = i - (size_t)cand;
            i += ml;
            lit = i;
            while (ml > 0) {
                const size_t l = ml > 64 ? (ml - 64 < 4 ? 60 : 64) : ml;
                if (l >= 4 && l <= 11 && off < 2048) {
                    out.push_back((uint8_t)(1 | ((l - 4) << 2) | ((off >> 8) << 5)));
                    out.push_back((uint8_t)off);
                } else {
                    out.push_back((uint8_t)(2 | ((l - 1) << 2)));
                    out.push_back((uint8_t)off);
                    out.push_back((uint8_t)(off >> 8));
                }
                ml -= l;
            }
        }
        snappy_literal(p + lit, len - lit, out);
    }
}

template <typename T>
void minmax_bytes(const void* vals, int64_t n, uint8_t* mn, uint8_t* mx) {
    const T* v = (const T*)vals;
    T lo = v[0], hi = v[0];
    for (int64_t i = 1; i < n; i++) { if (v[i] < lo) lo = v[i]; if (v[i] > hi) hi = v[i]; }
    memcpy(mn, &lo, sizeof(T));
    memcpy(mx, &hi, sizeof(T));
}

void write_page_header(pqw* w, const Col& c, size_t payload, size_t stored, int64_t num_values, int64_t nulls,
                       const uint8_t* mn, const uint8_t* mx, int mlen, std::vector<uint8_t>& out) {
    th::Writer t;
    t.struct_begin();
    t.i32(1, 0);  // DATA_PAGE
    t.i32(2, (int32_t)payload);  // uncompressed_page_size
    t.i32(3, (int32_t)stored);   // compressed_page_size
    t.struct_field_begin(5);
    t.i32(1, (int32_t)num_values);
    t.i32(2, 0);  // PLAIN
    t.i32(3, 3);  // RLE
    t.i32(4, 3);  // RLE
    if (w->stats_mode) {
        t.struct_field_begin(5);
        if (c.max_def > 0) t.i64(3, nulls);
        if (mlen > 0) { t.binary(5, mx, (size_t)mlen); t.binary(6, mn, (size_t)mlen); }
        t.struct_end();
    }
    t.struct_end();
    t.struct_end();
    out.insert(out.end(), t.buf.begin(), t.buf.end());
}

int emit_page(pqw* w, int col, const std::vector<uint8_t>& payload, int64_t num_values, int64_t nulls,
              const uint8_t* mn, const uint8_t* mx, int mlen) {
    if (payload.size() > 0x7fffffffu) return -1;
    size_t before = w->buf.size();
    if (w->codec == 1) {
        std::vector<uint8_t> comp;
        snappy_compress(payload.data(), payload.size(), comp);
        if (comp.size() > 0x7fffffffu) return -1;
        write_page_header(w, w->cols[col], payload.size(), comp.size(), num_values, nulls, mn, mx, mlen, w->buf);
        w->buf.insert(w->buf.end(), comp.begin(), comp.end());
    } else {
        write_page_header(w, w->cols[col], payload.size(), payload.size(), num_values, nulls, mn, mx, mlen, w->buf);
        w->buf.insert(w->buf.end(), payload.begin(), payload.end());
    }
    ChunkAcc& a = w->cur[col];
    a.used = true;
    a.num_values += num_values;
    a.total_size += (int64_t)(w->buf.size() - before);
    return 0;
}

}  // namespace

extern "C" {

pqw* pqw_create(int stats_mode) {
    pqw* w = new pqw();
    w->stats_mode = stats_mode;
    w->buf.insert(w->buf.end(), {'P', 'A', 'R', '1'});
    return w;
}
void pqw_destroy(pqw* w) { delete w; }
int pqw_set_codec(pqw* w, int codec) {
    if (!w || (codec != 0 && codec != 1) || !w->rgs.empty() || w->in_rg) return -1;
    w->codec = codec;
    return 0;
}

// Test helper: the compressor on its own.
int64_t pqw_snappy_compress(const uint8_t* in, int64_t n, uint8_t* out, int64_t cap) {
    std::vector<uint8_t> b;
    snappy_compress(in, (size_t)n, b);
    if ((int64_t)b.size() > cap) return -(int64_t)b.size();
    memcpy(out, b.data(), b.size());
    return (int64_t)b.size();
}

int pqw_reserve(pqw* w, int64_t bytes) {
    if (!w || bytes < 0) return -1;
    w->buf.reserve((size_t)bytes);
    return 0;
}

// rep_types[i] is the repetition (0 required, 1 optional, 2 repeated) of the i-th element of the
// dotted path, as in the generated Field's `Types` (fields.go:163, 176-188).
int pqw_add_column(pqw* w, const char* dotted_path, int phys_type, int converted_type,
                   const int32_t* rep_types, int n) {
    if (!w || w->in_rg || !w->rgs.empty()) return -1;
    Col c;
    c.path = dotted_path;
    size_t s = 0;
    for (;;) {
        size_t d = c.path.find('.', s);
        c.parts.push_back(c.path.substr(s, d == std::string::npos ? d : d - s));
        if (d == std::string::npos) break;
        s = d + 1;
    }
    if ((int)c.parts.size() != n) return -1;
    for (int i = 0; i < n; i++) {
        c.reps.push_back(rep_types[i]);
        c.max_def += rep_types[i] == 1 || rep_types[i] == 2;
        c.max_rep += rep_types[i] == 2;
    }
    c.phys = phys_type;
    c.conv = converted_type;
    w->cols.push_back(c);
    return (int)w->cols.size() - 1;
}

int pqw_begin_row_group(pqw* w) {
    if (!w || w->in_rg) return -1;
    w->in_rg = true;
    w->cur.assign(w->cols.size(), ChunkAcc());
    w->last_col = -1;
    return 0;
}

// One data page of column `col`.  Pages of one column must be written back to back and columns in
// increasing order (the reader consumes the file sequentially).
//   required column: defs = reps = NULL, n_levels == n_values
//   fixed width: values = n_values little-endian elements;  bool: n_values bytes (0/1);
//   string: values = chars, str_offsets[n_values + 1]
int pqw_write_page(pqw* w, int col, const uint8_t* defs, const uint8_t* reps, int64_t n_levels,
                   const void* values, int64_t n_values, const int32_t* str_offsets) {
    if (!w || !w->in_rg || col < 0 || (size_t)col >= w->cols.size() || col < w->last_col) return -1;
    w->last_col = col;
    const Col& c = w->cols[col];
    std::vector<uint8_t> payload;
    if (c.max_rep > 0) {
        if (!reps) return -1;
        LevelEncoder(bits_len((unsigned)c.max_rep)).encode(reps, (size_t)n_levels, payload);
    }
    if (c.max_def > 0) {
        if (!defs) return -1;
        LevelEncoder(bits_len((unsigned)c.max_def)).encode(defs, (size_t)n_levels, payload);
    }
    uint8_t mn[8], mx[8];
    int mlen = 0;
    const uint8_t* v = (const uint8_t*)values;
    switch (c.phys) {
    case 1: case 4:
        payload.insert(payload.end(), v, v + n_values * 4);
        if (w->stats_mode && n_values > 0) {
            mlen = 4;
            if (c.phys == 4) minmax_bytes<float>(values, n_values, mn, mx);
            else if (c.conv == 13) minmax_bytes<uint32_t>(values, n_values, mn, mx);
            else minmax_bytes<int32_t>(values, n_values, mn, mx);
        }
        break;
    case 2: case 5:
        payload.insert(payload.end(), v, v + n_values * 8);
        if (w->stats_mode && n_values > 0) {
            mlen = 8;
            if (c.phys == 5) minmax_bytes<double>(values, n_values, mn, mx);
            else if (c.conv == 14) minmax_bytes<uint64_t>(values, n_values, mn, mx);
            else minmax_bytes<int64_t>(values, n_values, mn, mx);
        }
        break;
    case 0: {
        size_t nb = (size_t)(n_values + 7) / 8, base = payload.size();
        payload.insert(payload.end(), nb, 0);
        for (int64_t i = 0; i < n_values; i++)
            if (v[i]) payload[base + (size_t)(i >> 3)] |= (uint8_t)(1u << (i & 7));
        break;
    }
    case 6: {
        if (!str_offsets) return -1;
        for (int64_t i = 0; i < n_values; i++) {
            uint32_t L = (uint32_t)(str_offsets[i + 1] - str_offsets[i]);
            uint8_t lb[4];
            memcpy(lb, &L, 4);
            payload.insert(payload.end(), lb, lb + 4);
            payload.insert(payload.end(), v + str_offsets[i], v + str_offsets[i + 1]);
        }
        break;
    }
    default:
        return -1;
    }
    int64_t nulls = c.max_def > 0 ? n_levels - n_values : 0;
    int64_t count = c.max_def > 0 ? n_levels : n_values;
    return emit_page(w, col, payload, count, nulls, mn, mx, mlen);
}

// A page whose payload bytes are supplied verbatim (hand-built level blocks, fixtures, malformed input).
int pqw_write_raw_page(pqw* w, int col, const uint8_t* payload, int64_t payload_len, int64_t num_values) {
    if (!w || !w->in_rg || col < 0 || (size_t)col >= w->cols.size() || col < w->last_col) return -1;
    w->last_col = col;
    std::vector<uint8_t> p(payload, payload + payload_len);
    return emit_page(w, col, p, num_values, 0, nullptr, nullptr, 0);
}

int pqw_end_row_group(pqw* w, int64_t num_rows) {
    if (!w || !w->in_rg) return -1;
    w->in_rg = false;
    pqw::RG rg;
    rg.rows = num_rows;
    rg.chunks = w->cur;
    w->rgs.push_back(rg);
    w->total_rows += num_rows;
    return 0;
}

int pqw_finish(pqw* w) {
    if (!w || w->in_rg || w->finished) return -1;
    th::Writer t;
    t.struct_begin();
    t.i32(1, 1);  // version
    // schema: root, then a group node per distinct path prefix, depth first in column order
    struct Node { std::string name; int rep; int children; int col; };
    std::vector<Node> nodes;
    nodes.push_back({"root", -1, 0, -1});
    std::map<std::string, size_t> seen;
    for (size_t ci = 0; ci < w->cols.size(); ci++) {
        const Col& c = w->cols[ci];
        std::string prefix;
        size_t parent = 0;
        for (size_t k = 0; k + 1 < c.parts.size(); k++) {
            prefix += (k ? "." : "") + c.parts[k];
            auto it = seen.find(prefix);
            if (it == seen.end()) {
                nodes[parent].children++;
                nodes.push_back({c.parts[k], c.reps[k], 0, -1});
                it = seen.emplace(prefix, nodes.size() - 1).first;
            }
            parent = it->second;
        }
        nodes[parent].children++;
        nodes.push_back({c.parts.back(), c.reps.back(), 0, (int)ci});
    }
    t.list_begin(2, th::T_STRUCT, nodes.size());
    for (const Node& n : nodes) {
        t.struct_begin();
        if (n.col >= 0) t.i32(1, w->cols[n.col].phys);
        if (n.rep >= 0) t.i32(3, n.rep);
        t.str(4, n.name);
        if (n.col < 0) t.i32(5, n.children);
        if (n.col >= 0 && w->cols[n.col].conv >= 0) t.i32(6, w->cols[n.col].conv);
        t.struct_end();
    }
    t.i64(3, w->total_rows);
    size_t n_rg = 0;
    for (const auto& rg : w->rgs) n_rg += rg.rows != 0;  // parquet.go:229-231
    t.list_begin(4, th::T_STRUCT, n_rg);
    int64_t pos = 4;
    for (const auto& rg : w->rgs) {
        if (rg.rows == 0) continue;
        size_t n_ch = 0;
        int64_t total = 0;
        for (const auto& ch : rg.chunks) { n_ch += ch.used; if (ch.used) total += ch.total_size; }
        t.struct_begin();
        t.list_begin(1, th::T_STRUCT, n_ch);
        for (size_t ci = 0; ci < rg.chunks.size(); ci++) {
            const ChunkAcc& ch = rg.chunks[ci];
            if (!ch.used) continue;
            const Col& c = w->cols[ci];
            t.struct_begin();               // ColumnChunk
            t.i64(2, pos);                  // file_offset
            t.struct_field_begin(3);        // ColumnMetaData
            t.i32(1, c.phys);
            t.list_begin(2, th::T_I32, 1);
            t.raw_i32(0);                   // encodings = [PLAIN]
            t.list_begin(3, th::T_BINARY, c.parts.size());
            for (const auto& p : c.parts) t.raw_binary(p.data(), p.size());
            t.i32(4, w->codec);             // CompressionCodec
            t.i64(5, ch.num_values);
            t.i64(6, ch.total_size);
            t.i64(7, ch.total_size);
            t.i64(9, pos);                  // data_page_offset
            t.struct_end();
            t.struct_end();
            pos += ch.total_size;
        }
        t.i64(2, total);
        t.i64(3, rg.rows);
        t.struct_end();
    }
    t.struct_end();
    w->buf.insert(w->buf.end(), t.buf.begin(), t.buf.end());
    uint32_t L = (uint32_t)t.buf.size();
    uint8_t lb[4];
    memcpy(lb, &L, 4);
    w->buf.insert(w->buf.end(), lb, lb + 4);
    w->buf.insert(w->buf.end(), {'P', 'A', 'R', '1'});
    w->finished = true;
    return 0;
}

const uint8_t* pqw_data(pqw* w, int64_t* len) {
    if (!w) return nullptr;
    if (len) *len = (int64_t)w->buf.size();
    return w->buf.data();
}

// Level block exactly as the writer emits it (for tests of the encoder on its own).
int64_t pqw_encode_levels(int width, const uint8_t* levels, int64_t n, uint8_t* out, int64_t cap) {
    std::vector<uint8_t> b;
    LevelEncoder(width).encode(levels, (size_t)n, b);
    if ((int64_t)b.size() > cap) return -(int64_t)b.size();
    memcpy(out, b.data(), b.size());
    return (int64_t)b.size();
}

}  // extern "C"
