// Host-side file walk: footer (thrift FileMetaData) + sequential page-header walk → the chunk and
// page descriptor tables the decoder consumes.
//
// Mirrors what the reference reader does before and between page decodes:
//   ReadMetaData / ReadFooter      parquet.go:349-370   (footer length at EOF-8, thrift compact)
//   Metadata.Pages / RowGroups     parquet.go:322-346, 142-151
//   readRowGroup                   cmd/parquetgen/gen/template.go:320-351 (chunks in rg.Columns order,
//                                  file consumed strictly sequentially from byte 4, template.go:263)
//   PageHeader                     parquet.go:373-378
//   page-loop termination          fields.go:118 (required: values) and fields.go:281,314 (optional: bytes)
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/pqdecode.h"
#include "thrift_compact.h"

using pqd::thrift::Reader;
namespace th = pqd::thrift;

namespace {

struct SchemaElem {
    std::string name;
    int type = -1, repetition = 0, num_children = 0, converted = -1;
    bool has_type = false;
};

struct Column {
    std::string path;
    pqd_column_info info;
};

struct ChunkMeta {
    int col = -1;
    int codec = 0;
    int64_t num_values = 0, total_compressed = 0, total_uncompressed = 0;
    int64_t data_page_offset = -1, dictionary_page_offset = -1, file_offset = -1;
    std::string path;
    // filled by the page walk
    int64_t first_page = 0, n_pages = 0;
    int64_t start_pos = 0;
};

struct RowGroup {
    int64_t num_rows = 0;
    std::vector<ChunkMeta> chunks;
};

struct PageHdr {
    int32_t type = 0, uncompressed = 0, compressed = 0, num_values = 0;
    int32_t encoding = 0, def_enc = 0, rep_enc = 0;
    bool has_dph = false;
    int32_t header_len = 0;
};

bool parse_page_header(const uint8_t* p, size_t len, PageHdr& h) {
    Reader r(p, len);
    r.enter_struct();
    int id, t, seen = 0;
    while (r.next_field(id, t)) {
        if (id == 1 && t == th::T_I32) { h.type = r.i32(); seen |= 1; }
        else if (id == 2 && t == th::T_I32) { h.uncompressed = r.i32(); seen |= 2; }
        else if (id == 3 && t == th::T_I32) { h.compressed = r.i32(); seen |= 4; }
        else if (id == 5 && t == th::T_STRUCT) {
            h.has_dph = true;
            r.enter_struct();
            int did, dt, dseen = 0;
            while (r.next_field(did, dt)) {
                if (did == 1 && dt == th::T_I32) { h.num_values = r.i32(); dseen |= 1; }
                else if (did == 2 && dt == th::T_I32) { h.encoding = r.i32(); dseen |= 2; }
                else if (did == 3 && dt == th::T_I32) { h.def_enc = r.i32(); dseen |= 4; }
                else if (did == 4 && dt == th::T_I32) { h.rep_enc = r.i32(); dseen |= 8; }
                else r.skip(dt);
            }
            if (!r.ok() || dseen != 15) return false;
        } else r.skip(t);
    }
    if (!r.ok() || seen != 7) return false;
    h.header_len = (int32_t)r.pos();
    return true;
}

}  // namespace

struct pqd_file {
    const uint8_t* buf = nullptr;
    size_t len = 0;
    int64_t num_rows = 0;
    std::vector<Column> columns;
    std::map<std::string, int> col_by_path;
    std::vector<RowGroup> row_groups;
    std::vector<pqd_page_desc> pages;      // all pages, in walk order
    std::vector<PageHdr> page_headers;
    std::vector<int64_t> page_header_pos;
    std::vector<pqd_chunk_desc> out_chunks; // scratch for pqd_file_descriptors
    std::vector<pqd_page_desc> out_pages;
};

namespace {

bool parse_schema(Reader& r, std::vector<SchemaElem>& out) {
    int et; uint32_t n;
    if (!r.list_header(et, n) || et != th::T_STRUCT) return false;
    for (uint32_t k = 0; k < n && r.ok(); k++) {
        SchemaElem e;
        r.enter_struct();
        int id, t;
        while (r.next_field(id, t)) {
            if (id == 1 && t == th::T_I32) { e.type = r.i32(); e.has_type = true; }
            else if (id == 3 && t == th::T_I32) e.repetition = r.i32();
            else if (id == 4 && t == th::T_BINARY) e.name = r.str();
            else if (id == 5 && t == th::T_I32) e.num_children = r.i32();
            else if (id == 6 && t == th::T_I32) e.converted = r.i32();
            else r.skip(t);
        }
        out.push_back(e);
    }
    return r.ok();
}

bool parse_column_meta(Reader& r, ChunkMeta& c) {
    r.enter_struct();
    int id, t;
    while (r.next_field(id, t)) {
        if (id == 3 && t == th::T_LIST) {
            int et; uint32_t n;
            r.list_header(et, n);
            for (uint32_t k = 0; k < n && r.ok(); k++) {
                if (k) c.path += ".";
                c.path += r.str();
            }
        } else if (id == 4 && t == th::T_I32) c.codec = r.i32();
        else if (id == 5 && t == th::T_I64) c.num_values = r.i64();
        else if (id == 6 && t == th::T_I64) c.total_uncompressed = r.i64();
        else if (id == 7 && t == th::T_I64) c.total_compressed = r.i64();
        else if (id == 9 && t == th::T_I64) c.data_page_offset = r.i64();
        else if (id == 11 && t == th::T_I64) c.dictionary_page_offset = r.i64();
        else r.skip(t);
    }
    return r.ok();
}

bool parse_row_group(Reader& r, RowGroup& rg) {
    r.enter_struct();
    int id, t;
    while (r.next_field(id, t)) {
        if (id == 1 && t == th::T_LIST) {
            int et; uint32_t n;
            r.list_header(et, n);
            for (uint32_t k = 0; k < n && r.ok(); k++) {
                ChunkMeta c;
                r.enter_struct();
                int cid, ct;
                while (r.next_field(cid, ct)) {
                    if (cid == 2 && ct == th::T_I64) c.file_offset = r.i64();
                    else if (cid == 3 && ct == th::T_STRUCT) { if (!parse_column_meta(r, c)) return false; }
                    else r.skip(ct);
                }
                rg.chunks.push_back(c);
            }
        } else if (id == 3 && t == th::T_I64) rg.num_rows = r.i64();
        else r.skip(t);
    }
    return r.ok();
}

// Depth-first flattening of the schema list into leaf columns with their max levels
// (RepetitionTypes.MaxDef / MaxRep, fields.go:38-57).
bool flatten(const std::vector<SchemaElem>& s, size_t& idx, const std::string& prefix, int def, int rep,
             std::vector<Column>& out, int depth) {
    if (idx >= s.size() || depth > 64) return false;
    const SchemaElem& e = s[idx++];
    int d = def + (e.repetition == 1 || e.repetition == 2);
    int rp = rep + (e.repetition == 2);
    std::string path = prefix.empty() ? e.name : prefix + "." + e.name;
    if (e.num_children > 0) {
        for (int k = 0; k < e.num_children; k++)
            if (!flatten(s, idx, path, d, rp, out, depth + 1)) return false;
        return true;
    }
    if (!e.has_type) return false;
    Column c;
    c.path = path;
    c.info.phys_type = e.type;
    c.info.converted_type = e.converted;
    c.info.max_def = d;
    c.info.max_rep = rp;
    c.info.leaf_repetition = e.repetition;
    c.info.reserved = 0;
    out.push_back(c);
    return true;
}

void set_err(char* err, size_t cap, const char* fmt, const char* a = "", long long b = 0) {
    if (err && cap) snprintf(err, cap, fmt, a, b);
}

}  // namespace

extern "C" {

int pqd_file_open(const void* buf_, size_t len, pqd_file** out, char* err, size_t err_cap) {
    if (!buf_ || !out) return PQD_E_ARG;
    const uint8_t* buf = (const uint8_t*)buf_;
    if (len < 12 || memcmp(buf, "PAR1", 4) != 0 || memcmp(buf + len - 4, "PAR1", 4) != 0) {
        set_err(err, err_cap, "not a parquet file%s", "");
        return PQD_E_FORMAT;
    }
    uint32_t flen;
    memcpy(&flen, buf + len - 8, 4);  // getMetaDataSize, parquet.go:507-515
    if ((size_t)flen + 12 > len) {
        set_err(err, err_cap, "footer length %s%lld exceeds file", "", (long long)flen);
        return PQD_E_FORMAT;
    }
    pqd_file* f = new pqd_file();
    f->buf = buf;
    f->len = len;
    Reader r(buf + len - 8 - flen, flen);
    std::vector<SchemaElem> schema;
    r.enter_struct();
    int id, t;
    bool ok = true;
    while (ok && r.next_field(id, t)) {
        if (id == 2 && t == th::T_LIST) ok = parse_schema(r, schema);
        else if (id == 3 && t == th::T_I64) f->num_rows = r.i64();
        else if (id == 4 && t == th::T_LIST) {
            int et; uint32_t n;
            r.list_header(et, n);
            for (uint32_t k = 0; k < n && ok; k++) {
                RowGroup rg;
                ok = parse_row_group(r, rg);
                f->row_groups.push_back(rg);
            }
        } else r.skip(t);
    }
    if (!ok || !r.ok() || schema.empty()) {
        set_err(err, err_cap, "thrift FileMetaData decode failed%s", "");
        delete f;
        return PQD_E_FORMAT;
    }
    // leaves
    size_t idx = 1;
    for (int k = 0; k < schema[0].num_children; k++) {
        if (!flatten(schema, idx, "", 0, 0, f->columns, 0)) {
            set_err(err, err_cap, "malformed schema tree%s", "");
            delete f;
            return PQD_E_FORMAT;
        }
    }
    for (size_t c = 0; c < f->columns.size(); c++) f->col_by_path[f->columns[c].path] = (int)c;

    // Sequential walk from byte 4 (template.go:263), chunks in rg.Columns() order (:332).
    size_t pos = 4;
    for (size_t g = 0; g < f->row_groups.size(); g++) {
        RowGroup& rg = f->row_groups[g];
        for (size_t c = 0; c < rg.chunks.size(); c++) {
            ChunkMeta& ch = rg.chunks[c];
            auto it = f->col_by_path.find(ch.path);
            if (it == f->col_by_path.end()) {  // "unknown field", template.go:335-337
                set_err(err, err_cap, "unknown field: %s", ch.path.c_str());
                delete f;
                return PQD_E_FORMAT;
            }
            ch.col = it->second;
            ch.first_page = (int64_t)f->pages.size();
            ch.start_pos = (int64_t)pos;
            const bool optional = f->columns[ch.col].info.max_def > 0;
            int64_t n_read = 0;
            const int64_t limit = optional ? ch.total_compressed : ch.num_values;
            while (n_read < limit) {
                PageHdr h;
                if (pos >= len - 8 - flen || !parse_page_header(buf + pos, len - pos, h)) {
                    set_err(err, err_cap, "unable to read page header of %s at %lld", ch.path.c_str(), (long long)pos);
                    delete f;
                    return PQD_E_FORMAT;
                }
                // pageData: the UNCOMPRESSED arm reads UncompressedPageSize bytes, the codecs read
                // CompressedPageSize (fields.go:362,374,393)
                int32_t body = ch.codec == 0 ? h.uncompressed : h.compressed;
                if (body < 0 || (size_t)body > len - pos - (size_t)h.header_len) {
                    set_err(err, err_cap, "page of %s at %lld runs past the end of the file", ch.path.c_str(), (long long)pos);
                    delete f;
                    return PQD_E_CORRUPT;
                }
                pqd_page_desc pd;
                pd.src_off = (int64_t)pos + h.header_len;
                pd.payload_len = body;
                pd.num_values = h.has_dph ? h.num_values : -1;
                pd.rep_bytes = -1;
                pd.def_bytes = -1;
                f->pages.push_back(pd);
                f->page_headers.push_back(h);
                f->page_header_pos.push_back((int64_t)pos);
                pos += (size_t)h.header_len + (size_t)body;
                n_read += optional ? (int64_t)h.header_len + body : (int64_t)(h.has_dph ? h.num_values : 0);
                if (!h.has_dph && !optional) break;  // the reference would dereference nil here
            }
            ch.n_pages = (int64_t)f->pages.size() - ch.first_page;
        }
    }
    *out = f;
    return PQD_OK;
}

void pqd_file_close(pqd_file* f) { delete f; }
int64_t pqd_file_num_rows(const pqd_file* f) { return f ? f->num_rows : 0; }
int32_t pqd_file_num_columns(const pqd_file* f) { return f ? (int32_t)f->columns.size() : 0; }
int32_t pqd_file_num_row_groups(const pqd_file* f) { return f ? (int32_t)f->row_groups.size() : 0; }
int64_t pqd_file_row_group_rows(const pqd_file* f, int32_t rg) {
    if (!f || rg < 0 || (size_t)rg >= f->row_groups.size()) return -1;
    return f->row_groups[rg].num_rows;
}
int pqd_file_column(const pqd_file* f, int32_t col, pqd_column_info* info) {
    if (!f || !info || col < 0 || (size_t)col >= f->columns.size()) return PQD_E_ARG;
    *info = f->columns[col].info;
    return PQD_OK;
}
const char* pqd_file_column_path(const pqd_file* f, int32_t col) {
    if (!f || col < 0 || (size_t)col >= f->columns.size()) return nullptr;
    return f->columns[col].path.c_str();
}
int64_t pqd_file_num_pages(const pqd_file* f) { return f ? (int64_t)f->pages.size() : 0; }

int pqd_file_page_header(const pqd_file* f, int64_t page, int32_t out[8], int64_t* header_pos) {
    if (!f || page < 0 || (size_t)page >= f->pages.size() || !out) return PQD_E_ARG;
    const PageHdr& h = f->page_headers[page];
    out[0] = h.type; out[1] = h.uncompressed; out[2] = h.compressed; out[3] = h.num_values;
    out[4] = h.encoding; out[5] = h.def_enc; out[6] = h.rep_enc; out[7] = h.header_len;
    if (header_pos) *header_pos = f->page_header_pos[page];
    return PQD_OK;
}

int pqd_file_chunk_codec(const pqd_file* f, int32_t rg, int32_t chunk) {
    if (!f || rg < 0 || (size_t)rg >= f->row_groups.size()) return PQD_E_ARG;
    const RowGroup& g = f->row_groups[rg];
    if (chunk < 0 || (size_t)chunk >= g.chunks.size()) return PQD_E_ARG;
    return g.chunks[chunk].codec;
}

int pqd_file_chunk_meta(const pqd_file* f, int32_t rg, int32_t chunk, int64_t out[6]) {
    if (!f || !out || rg < 0 || (size_t)rg >= f->row_groups.size()) return PQD_E_ARG;
    const RowGroup& g = f->row_groups[rg];
    if (chunk < 0 || (size_t)chunk >= g.chunks.size()) return PQD_E_ARG;
    const ChunkMeta& c = g.chunks[chunk];
    out[0] = c.col; out[1] = c.num_values; out[2] = c.total_compressed; out[3] = c.start_pos;
    out[4] = c.first_page; out[5] = c.n_pages;
    return PQD_OK;
}

int pqd_file_descriptors(pqd_file* f, int32_t rg_first, int32_t rg_count,
                         const pqd_chunk_desc** chunks, size_t* n_chunks,
                         const pqd_page_desc** pages, size_t* n_pages) {
    if (!f || rg_first < 0 || rg_count < 0 || (size_t)rg_first + (size_t)rg_count > f->row_groups.size())
        return PQD_E_ARG;
    f->out_chunks.clear();
    f->out_pages.clear();
    for (int32_t g = rg_first; g < rg_first + rg_count; g++) {
        for (const ChunkMeta& ch : f->row_groups[g].chunks) {
            if ((ch.codec != 0 && ch.codec != 1) || ch.dictionary_page_offset > 0) return PQD_E_UNSUPPORTED;
            const pqd_column_info& ci = f->columns[ch.col].info;
            pqd_chunk_desc cd;
            cd.phys_type = ci.phys_type;
            cd.max_def = ci.max_def;
            cd.max_rep = ci.max_rep;
            cd.first_page = (int32_t)f->out_pages.size();
            cd.n_pages = (int32_t)ch.n_pages;
            cd.column = ch.col;
            cd.num_values = ch.num_values;
            cd.codec = ch.codec;
            cd.reserved = 0;
            f->out_chunks.push_back(cd);
            for (int64_t p = 0; p < ch.n_pages; p++) {
                const PageHdr& h = f->page_headers[ch.first_page + p];
                if (h.type != 0 || !h.has_dph || h.encoding != 0) return PQD_E_UNSUPPORTED;  // DATA_PAGE + PLAIN only
                f->out_pages.push_back(f->pages[ch.first_page + p]);
            }
        }
    }
    if (chunks) *chunks = f->out_chunks.data();
    if (n_chunks) *n_chunks = f->out_chunks.size();
    if (pages) *pages = f->out_pages.data();
    if (n_pages) *n_pages = f->out_pages.size();
    return PQD_OK;
}

}  // extern "C"
