// C++ host-side mirror of the reader that parquetgen generates (cmd/parquetgen/gen/template.go:228-382),
// written over the C ABI of pqdecode.h.  Same names, same meaning, same error behaviour:
//
//   NewParquetReader(r)   template.go:235-270   reads the footer, walks the page headers, reads the first row group
//   Rows()                template.go:353-355   FileMetaData.num_rows
//   Next()                template.go:357-371   advances one row, reading the next row group when the current one is
//                                               used up; false at the end or after an error
//   Scan(&x)              template.go:373-382   hands the current row to every field; here a field yields the slices
//                                               of values and levels that the generated write function of that field
//                                               consumes for the row (write_repeated.go:62-80, template_optional.go)
//   Error()               template.go:316-318
//   Levels()              template.go:301-314   definition / repetition levels of the row group held
//
// A row group is decoded in one call (readRowGroup, template.go:320-351: every Field.Read of the group) on
// the device; Next and Scan then only index host arrays.  Header only; link with -lpqdecode.
#ifndef PQREADER_HPP
#define PQREADER_HPP

#include <cstdint>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "pqdecode.h"

namespace pq {

// One leaf column of the row group held by the reader: what the generated Field keeps after Read
// (fields.go:38-57 RequiredField, :147-160 OptionalField: vals, Defs, Reps).
struct Field {
    std::string name;  // dotted path, Field.Name()
    int phys_type = 0, max_def = 0, max_rep = 0;
    int64_t n_levels = 0, n_values = 0, n_rows = 0;
    std::vector<uint8_t> defs, reps;   // empty for required / non-repeated columns
    std::vector<uint8_t> values;       // dense little-endian values; one byte per bool; unused for strings
    std::vector<int32_t> offsets;      // strings: n_values + 1 offsets into chars
    std::vector<uint8_t> chars;
    std::vector<uint8_t> validity;     // bit i (LSB first): level i holds a value
    std::vector<int32_t> row_level_off, row_value_off;  // repeated columns: first level / value of every row
    int value_width() const {
        switch (phys_type) {
        case PQD_INT32: case PQD_FLOAT: return 4;
        case PQD_INT64: case PQD_DOUBLE: return 8;
        case PQD_BOOLEAN: return 1;
        default: return 0;
        }
    }
};

// Levels (template.go:295-299)
struct Levels {
    std::string name;
    const std::vector<uint8_t>* defs;
    const std::vector<uint8_t>* reps;
};

// What one field contributes to the current row.
struct Cell {
    const Field* field = nullptr;
    int64_t first_value = 0, n_values = 0;  // indices into field->values (× value_width) or field->offsets
    int64_t first_level = 0, n_levels = 0;  // indices into field->defs / field->reps (0 levels for required columns)
    const void* values() const { return field->values.data() + first_value * field->value_width(); }
    const uint8_t* defs() const { return field->defs.empty() ? nullptr : field->defs.data() + first_level; }
    const uint8_t* reps() const { return field->reps.empty() ? nullptr : field->reps.data() + first_level; }
    // string number i of the cell
    std::string str(int64_t i) const {
        const int32_t a = field->offsets[first_value + i], b = field->offsets[first_value + i + 1];
        return std::string((const char*)field->chars.data() + a, (size_t)(b - a));
    }
};

class ParquetReader {
public:
    // NewParquetReader: the file bytes stay owned by the caller and must outlive the reader.  On failure
    // returns null and *err holds the message (the reference returns (nil, err)).
    static std::unique_ptr<ParquetReader> New(const void* buf, size_t len, int device, std::string* err) {
        std::unique_ptr<ParquetReader> r(new ParquetReader());
        r->buf_ = buf;
        r->len_ = len;
        char msg[256] = {0};
        if (pqd_file_open(buf, len, &r->file_, msg, sizeof msg) != PQD_OK) {
            if (err) *err = msg;
            return nullptr;
        }
        const int rc = pqd_create(device, &r->ctx_);
        if (rc != PQD_OK) {
            if (err) *err = "no usable device (status " + std::to_string(rc) + ")";
            return nullptr;
        }
        r->rows_ = pqd_file_num_rows(r->file_);
        r->n_groups_ = pqd_file_num_row_groups(r->file_);
        const int32_t nc = pqd_file_num_columns(r->file_);
        for (int32_t c = 0; c < nc; c++) r->names_.push_back(pqd_file_column_path(r->file_, c));
        if (!r->readRowGroup()) {  // NewParquetReader fails when the first row group cannot be read
            if (err) *err = r->err_;
            return nullptr;
        }
        return r;
    }
    ~ParquetReader() {
        if (ctx_) pqd_destroy(ctx_);
        if (file_) pqd_file_close(file_);
    }
    ParquetReader(const ParquetReader&) = delete;
    ParquetReader& operator=(const ParquetReader&) = delete;

    int64_t Rows() const { return rows_; }
    const std::string& Error() const { return err_; }

    bool Next() {
        if (err_.empty() && cursor_ >= rows_) return false;
        if (!err_.empty()) return false;
        if (rg_cursor_ >= rg_count_) {
            if (!readRowGroup()) return false;
        }
        cursor_++;
        rg_cursor_++;
        return true;
    }

    // Scan: one Cell per field, in Fields() order, for the row Next() moved to.  After an error nothing is
    // scanned (template.go:374-376).
    void Scan(std::vector<Cell>& row) {
        row.clear();
        if (!err_.empty()) return;
        for (size_t i = 0; i < fields_.size(); i++) {
            const Field& f = fields_[i];
            const int64_t r = row_pos_[i]++;
            Cell c;
            c.field = &f;
            if (f.max_def == 0) {
                c.first_value = r; c.n_values = 1;
            } else if (f.max_rep == 0) {  // row r is level r; its value index is the number of values before it
                c.first_level = r; c.n_levels = 1;
                c.first_value = value_cursor_[i];
                c.n_values = (f.validity[(size_t)r >> 3] >> (r & 7)) & 1;
                value_cursor_[i] += c.n_values;
            } else {
                c.first_level = f.row_level_off[(size_t)r]; c.n_levels = f.row_level_off[(size_t)r + 1] - c.first_level;
                c.first_value = f.row_value_off[(size_t)r]; c.n_values = f.row_value_off[(size_t)r + 1] - c.first_value;
            }
            row.push_back(c);
        }
    }

    const std::vector<Field>& Fields() const { return fields_; }
    std::vector<Levels> GetLevels() const {
        std::vector<Levels> out;
        for (const Field& f : fields_) out.push_back(Levels{f.name, &f.defs, &f.reps});
        return out;
    }

private:
    ParquetReader() = default;

    template <class T>
    bool fetch(pqd_plan* plan, size_t chunk, pqd_output which, std::vector<T>& dst, size_t n) {
        dst.assign(n, T());
        if (n == 0) return true;
        const int64_t got = pqd_plan_fetch(plan, chunk, which, dst.data(), n * sizeof(T));
        if (got < 0) return fail((int)got);
        return true;
    }
    bool fail(int rc) {
        const char* m = ctx_ ? pqd_last_error(ctx_) : nullptr;
        err_ = (m && *m) ? std::string(m) : "status " + std::to_string(rc);
        return false;
    }

    // readRowGroup (template.go:320-351)
    bool readRowGroup() {
        rg_cursor_ = 0;
        if (rg_next_ >= n_groups_) { rg_count_ = 0; return true; }
        const pqd_chunk_desc* chunks = nullptr;
        const pqd_page_desc* pages = nullptr;
        size_t nc = 0, np = 0;
        int rc = pqd_file_descriptors(file_, rg_next_, 1, &chunks, &nc, &pages, &np);
        if (rc != PQD_OK) return fail(rc);
        pqd_plan* plan = nullptr;
        rc = pqd_decode(ctx_, chunks, nc, pages, np, buf_, len_, &plan);
        if (rc != PQD_OK) {
            fail(rc);
            if (plan) pqd_plan_destroy(plan);
            return false;
        }
        fields_.assign(nc, Field());
        bool ok = true;
        for (size_t c = 0; c < nc && ok; c++) {
            Field& f = fields_[c];
            pqd_chunk_info ci;
            if ((rc = pqd_plan_chunk_info(plan, c, &ci)) != PQD_OK) { ok = fail(rc); break; }
            f.name = names_[(size_t)chunks[c].column];
            f.phys_type = chunks[c].phys_type; f.max_def = chunks[c].max_def; f.max_rep = chunks[c].max_rep;
            f.n_levels = ci.n_levels; f.n_values = ci.n_values; f.n_rows = ci.n_rows;
            if (f.max_def > 0) {
                ok = ok && fetch(plan, c, PQD_OUT_DEFS, f.defs, (size_t)ci.n_levels);
                ok = ok && fetch(plan, c, PQD_OUT_VALIDITY, f.validity, ((size_t)ci.n_levels + 7) / 8);
            }
            if (f.max_rep > 0) {
                ok = ok && fetch(plan, c, PQD_OUT_REPS, f.reps, (size_t)ci.n_levels);
                ok = ok && fetch(plan, c, PQD_OUT_ROW_LEVEL_OFF, f.row_level_off, (size_t)ci.n_rows + 1);
                ok = ok && fetch(plan, c, PQD_OUT_ROW_VALUE_OFF, f.row_value_off, (size_t)ci.n_rows + 1);
            }
            if (f.phys_type == PQD_BYTE_ARRAY) {
                ok = ok && fetch(plan, c, PQD_OUT_OFFSETS, f.offsets, (size_t)ci.n_values + 1);
                ok = ok && fetch(plan, c, PQD_OUT_CHARS, f.chars, (size_t)ci.chars_bytes);
            } else {
                ok = ok && fetch(plan, c, PQD_OUT_VALUES, f.values, (size_t)ci.value_bytes);
            }
        }
        pqd_plan_destroy(plan);
        if (!ok) return false;
        row_pos_.assign(nc, 0);
        value_cursor_.assign(nc, 0);
        rg_count_ = pqd_file_row_group_rows(file_, rg_next_);
        rg_next_++;
        return true;
    }

    const void* buf_ = nullptr;
    size_t len_ = 0;
    pqd_file* file_ = nullptr;
    pqd_ctx* ctx_ = nullptr;
    int64_t rows_ = 0, cursor_ = 0, rg_cursor_ = 0, rg_count_ = 0;
    int32_t n_groups_ = 0, rg_next_ = 0;
    std::string err_;
    std::vector<std::string> names_;
    std::vector<Field> fields_;
    std::vector<int64_t> row_pos_, value_cursor_;
};

}  // namespace pq

#endif
