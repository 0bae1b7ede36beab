// Thrift compact protocol: the subset Parquet metadata needs (reader and writer).
//
// The reference reads FileMetaData and PageHeader through github.com/apache/thrift v0.18.1's
// TCompactProtocol (parquet.go:349-378); that library is not part of the reference tree, so the
// wire format is implemented here from its published description:
//   field header  = (field-id delta << 4) | type, delta 0 → zigzag varint field id follows
//   i16/i32/i64   = zigzag varint;  binary/string = varint length + bytes;  double = 8 bytes LE
//   list/set      = (size << 4) | elem type, size 15 → varint size follows
//   struct        = fields then 0x00;  bool fields carry their value in the type nibble (1/2)
#pragma once
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace pqd {
namespace thrift {

enum CType : uint8_t {
    T_STOP = 0, T_TRUE = 1, T_FALSE = 2, T_BYTE = 3, T_I16 = 4, T_I32 = 5, T_I64 = 6,
    T_DOUBLE = 7, T_BINARY = 8, T_LIST = 9, T_SET = 10, T_MAP = 11, T_STRUCT = 12
};

class Reader {
  public:
    Reader(const uint8_t* p, size_t len) : p_(p), len_(len) {}
    bool ok() const { return ok_; }
    size_t pos() const { return i_; }

    // Reads the next field header; returns false on STOP or error.
    bool next_field(int& id, int& type) {
        uint8_t h = byte();
        if (!ok_ || h == 0) { leave_struct(); return false; }
        type = h & 0x0F;
        int delta = h >> 4;
        if (delta == 0) id = (int)zz(varint());
        else id = last_.back() + delta;
        last_.back() = id;
        return ok_;
    }
    void enter_struct() { last_.push_back(0); }
    int32_t i32() { return (int32_t)zz(varint()); }
    int64_t i64() { return zz(varint()); }
    bool list_header(int& elem_type, uint32_t& n) {
        uint8_t b = byte();
        n = b >> 4;
        elem_type = b & 0x0F;
        if (n == 15) n = (uint32_t)varint();
        return ok_;
    }
    std::string str() {
        uint64_t n = varint();
        if (!ok_ || n > len_ - i_) { ok_ = false; return std::string(); }
        std::string s((const char*)p_ + i_, (size_t)n);
        i_ += (size_t)n;
        return s;
    }
    void skip(int type, int depth = 0) {
        if (!ok_ || depth > 32) { ok_ = false; return; }
        switch (type) {
        case T_TRUE: case T_FALSE: break;
        case T_BYTE: byte(); break;
        case T_I16: case T_I32: case T_I64: varint(); break;
        case T_DOUBLE: if (len_ - i_ < 8) ok_ = false; else i_ += 8; break;
        case T_BINARY: {
            uint64_t n = varint();
            if (!ok_ || n > len_ - i_) ok_ = false; else i_ += (size_t)n;
            break;
        }
        case T_LIST: case T_SET: {
            int et; uint32_t n;
            list_header(et, n);
            for (uint32_t k = 0; ok_ && k < n; k++) {
                if (et == T_TRUE || et == T_FALSE) byte(); else skip(et, depth + 1);
            }
            break;
        }
        case T_MAP: {
            uint64_t n = varint();
            if (n == 0) break;
            uint8_t kv = byte();
            for (uint64_t k = 0; ok_ && k < n; k++) { skip(kv >> 4, depth + 1); skip(kv & 0x0F, depth + 1); }
            break;
        }
        case T_STRUCT: {
            enter_struct();
            int id, t;
            while (next_field(id, t)) skip(t, depth + 1);
            break;
        }
        default: ok_ = false;
        }
    }

  private:
    void leave_struct() { if (!last_.empty()) last_.pop_back(); }
    uint8_t byte() {
        if (i_ >= len_) { ok_ = false; return 0; }
        return p_[i_++];
    }
    uint64_t varint() {
        uint64_t r = 0;
        for (int shift = 0; shift < 70; shift += 7) {
            uint8_t b = byte();
            if (!ok_) return 0;
            r |= (uint64_t)(b & 0x7F) << shift;
            if (!(b & 0x80)) return r;
        }
        ok_ = false;
        return 0;
    }
    static int64_t zz(uint64_t v) { return (int64_t)(v >> 1) ^ -(int64_t)(v & 1); }

    const uint8_t* p_;
    size_t len_, i_ = 0;
    bool ok_ = true;
    std::vector<int> last_;
};

class Writer {
  public:
    std::vector<uint8_t> buf;
    void struct_begin() { last_.push_back(0); }
    void struct_end() { buf.push_back(0); last_.pop_back(); }
    void field(int id, int type) {
        int delta = id - last_.back();
        if (delta > 0 && delta <= 15) buf.push_back((uint8_t)((delta << 4) | type));
        else { buf.push_back((uint8_t)type); varint(zz((int64_t)id)); }
        last_.back() = id;
    }
    void i32(int id, int32_t v) { field(id, T_I32); varint(zz(v)); }
    void i64(int id, int64_t v) { field(id, T_I64); varint(zz(v)); }
    void binary(int id, const void* p, size_t n) { field(id, T_BINARY); raw_binary(p, n); }
    void str(int id, const std::string& s) { binary(id, s.data(), s.size()); }
    void list_begin(int id, int elem_type, size_t n) {
        field(id, T_LIST);
        if (n < 15) buf.push_back((uint8_t)((n << 4) | elem_type));
        else { buf.push_back((uint8_t)(0xF0 | elem_type)); varint(n); }
    }
    void struct_field_begin(int id) { field(id, T_STRUCT); struct_begin(); }
    void raw_i32(int32_t v) { varint(zz(v)); }
    void raw_binary(const void* p, size_t n) {
        varint(n);
        const uint8_t* b = (const uint8_t*)p;
        buf.insert(buf.end(), b, b + n);
    }

  private:
    static uint64_t zz(int64_t v) { return ((uint64_t)v << 1) ^ (uint64_t)(v >> 63); }
    void varint(uint64_t v) {
        while (v >= 0x80) { buf.push_back((uint8_t)(v | 0x80)); v >>= 7; }
        buf.push_back((uint8_t)v);
    }
    std::vector<int> last_;
};

}  // namespace thrift
}  // namespace pqd
