// Reads a Parquet file with the C++ mirror of the generated reader (include/pqreader.hpp) the way a program
// written against the reference does:
//     r, err := NewParquetReader(f); for r.Next() { r.Scan(&row) }; err = r.Error()
// and prints, per column, what it saw: rows, values, levels and an FNV-1a hash over everything Scan handed
// out, row by row.  tests/test_gpu_cpp_reader.py compares that with the oracle.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "pqreader.hpp"

static void fnv(uint64_t& h, const void* p, size_t n) {
    const unsigned char* b = (const unsigned char*)p;
    for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; }
}

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s file.parquet [device]\n", argv[0]); return 2; }
    FILE* fp = fopen(argv[1], "rb");
    if (!fp) { perror(argv[1]); return 2; }
    fseek(fp, 0, SEEK_END);
    const long n = ftell(fp);
    fseek(fp, 0, SEEK_SET);
    std::vector<unsigned char> buf((size_t)n);
    if (n && fread(buf.data(), 1, (size_t)n, fp) != (size_t)n) { fprintf(stderr, "short read\n"); return 2; }
    fclose(fp);

    std::string err;
    auto r = pq::ParquetReader::New(buf.data(), buf.size(), argc > 2 ? atoi(argv[2]) : 0, &err);
    if (!r) { printf("error %s\n", err.c_str()); return 1; }
    const size_t nf = r->Fields().size();
    std::vector<std::string> names;
    for (const pq::Field& f : r->Fields()) names.push_back(f.name);
    std::vector<uint64_t> hash(nf, 1469598103934665603ull), values(nf, 0), levels(nf, 0);
    std::vector<pq::Cell> row;
    int64_t rows = 0;
    while (r->Next()) {
        r->Scan(row);
        for (size_t i = 0; i < row.size(); i++) {
            const pq::Cell& c = row[i];
            values[i] += (uint64_t)c.n_values;
            levels[i] += (uint64_t)c.n_levels;
            if (c.field->phys_type == PQD_BYTE_ARRAY) {
                for (int64_t k = 0; k < c.n_values; k++) {
                    const std::string s = c.str(k);
                    const uint32_t len = (uint32_t)s.size();
                    fnv(hash[i], &len, 4);
                    fnv(hash[i], s.data(), s.size());
                }
            } else {
                fnv(hash[i], c.values(), (size_t)c.n_values * (size_t)c.field->value_width());
            }
            if (c.defs()) fnv(hash[i], c.defs(), (size_t)c.n_levels);
            if (c.reps()) fnv(hash[i], c.reps(), (size_t)c.n_levels);
        }
        rows++;
    }
    if (!r->Error().empty()) { printf("error %s\n", r->Error().c_str()); return 1; }
    printf("rows %lld of %lld\n", (long long)rows, (long long)r->Rows());
    for (size_t i = 0; i < nf; i++)
        printf("%s %llu %llu %016llx\n", names[i].c_str(), (unsigned long long)values[i], (unsigned long long)levels[i],
               (unsigned long long)hash[i]);
    return 0;
}
