// host_sanitize.cpp -- the host-only code of libsa_hip (CSV extractor threads, row copying, record retrieval, the
// opt-in host index) under AddressSanitizer + UndefinedBehaviorSanitizer, on the CPU (sanitizers are not available
// on the GPU pool, and this code never touches the device):
//     g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all -D__HIP_PLATFORM_AMD__ \
//         -I/opt/rocm/include -Isuffixarray_amd/csrc tools/host_sanitize.cpp -o /tmp/host_sanitize -lpthread
// tests/test_host_cpu.py::test_host_code_under_sanitizers builds and runs it; exit code 0 = clean.
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "csv_ingest.hpp"
#include "records.hpp"
#include "host_index.hpp"

using namespace sa;

#define REQUIRE(c)                                                                  \
    do {                                                                            \
        if (!(c)) { fprintf(stderr, "host_sanitize: %s:%d: %s\n", __FILE__, __LINE__, #c); return 1; } \
    } while (0)

static int check_csv(const char* dir) {
    const std::string path = std::string(dir) + "/sanitize.csv";
    REQUIRE(synth_csv(path.c_str(), 40000, 7) == 0);
    sa_hip_csv_column ref{};
    setenv("SA_HIP_CSV_THREADS", "1", 1);
    REQUIRE(csv_extract_column(path.c_str(), "company_name", &ref) == 0);
    REQUIRE(ref.num_rows == 40000 && ref.text_len > 0 && ref.num_columns == 3 && ref.column_index == 1);
    for (const char* threads : {"2", "3", "8", "64"}) {
        setenv("SA_HIP_CSV_THREADS", threads, 1);
        sa_hip_csv_column c{};
        REQUIRE(csv_extract_column(path.c_str(), "company_name", &c) == 0);
        REQUIRE(c.num_rows == ref.num_rows && c.text_len == ref.text_len);
        REQUIRE(memcmp(c.text, ref.text, ref.text_len) == 0);
        REQUIRE(memcmp(c.row_text_starts, ref.row_text_starts, ref.num_rows * 8) == 0);
        REQUIRE(memcmp(c.row_file_offsets, ref.row_file_offsets, (ref.num_rows + 1) * 8) == 0);
        csv_free(&c);
    }
    sa_hip_csv_column none{};
    REQUIRE(csv_extract_column(path.c_str(), "no_such_column", &none) != 0);
    REQUIRE(csv_extract_column((path + ".missing").c_str(), "company_name", &none) != 0);
    // rows out of the file: the first, a middle one and the last
    FILE* f = fopen(path.c_str(), "rb");
    REQUIRE(f != nullptr);
    std::vector<u8> file;
    u8 buf[65536];
    size_t got;
    while ((got = fread(buf, 1, sizeof buf, f)) > 0) file.insert(file.end(), buf, buf + got);
    fclose(f);
    for (u64 r : {(u64)0, (u64)12345, ref.num_rows - 1}) {
        char* s = dup_row(file.data(), ref.row_file_offsets[r], ref.row_file_offsets[r + 1]);
        REQUIRE(s != nullptr && strlen(s) > 4 && strchr(s, '\n') == nullptr);
        free(s);
    }
    // hits -> distinct rows over a fake suffix array: every position of the column, in order
    HostU64Array starts;   // (what the handle keeps: copied here, adopted from the extractor by the CSV index)
    starts.assign(ref.row_text_starts, ref.row_text_starts + ref.num_rows);
    std::vector<u32> fake((size_t)ref.text_len);
    for (size_t i = 0; i < fake.size(); ++i) fake[i] = (u32)((i * 7919u) % fake.size());
    auto fetch = [&](u64 pos, u64 count, u32* out) { memcpy(out, fake.data() + pos, (size_t)count * 4); return 0; };
    std::vector<u64> rows;
    sa_hip_pair_u32 rg = {10, (u32)fake.size() - 5};
    for (u32 k : {1u, 17u, 1000u, 50000u}) {
        REQUIRE(distinct_rows(starts, rg, k, nullptr, 0, fetch, rows) == 0);
        REQUIRE(rows.size() == (k < 40000u ? k : 40000u) || rows.size() <= 40000u);
        for (u64 r : rows) REQUIRE(r < ref.num_rows);
    }
    sa_hip_pair_u32 miss = {5, 4};
    REQUIRE(distinct_rows(starts, miss, 10, nullptr, 0, fetch, rows) == 0 && rows.empty());
    csv_free(&ref);
    remove(path.c_str());
    return 0;
}

static int check_host_index() {
    std::vector<std::string> texts = {"banana", "mississippi", "a", "", "aaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaa", "abababababababababab",
                                      std::string("ab\0ab\0\0abab\0", 12)};
    std::string big;
    u64 s = 12345;
    for (int i = 0; i < 20000; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; big.push_back((char)('a' + (s >> 33) % 4)); }
    texts.push_back(big);
    for (const std::string& t : texts) {
        for (u32 L : {0u, 1u, 2u, 3u, 5u, 8u, 32u}) {
            HostIndex h;
            h.n_max = t.size();
            h.set_text(reinterpret_cast<const u8*>(t.data()), t.size());
            h.build(L);
            REQUIRE(h.verify() == 0);
            for (const char* q : {"a", "an", "ssi", "zz", "", "ab", "abab"}) {
                const sa_hip_pair_u32 r = h.query(reinterpret_cast<const u8*>(q), strlen(q));
                if (r.first != 0xFFFFFFFFu && (u32)(r.second - r.first + 1u) != 0u) {
                    REQUIRE(r.second < t.size());
                    const u64 c = (L && strlen(q) > L) ? L : strlen(q);
                    for (u32 j = r.first; j <= r.second; ++j) REQUIRE(t.compare(h.sa[j], c, q, c) == 0);
                }
            }
        }
    }
    return 0;
}

int main(int argc, char** argv) {
    const char* dir = argc > 1 ? argv[1] : "/tmp";
    if (check_csv(dir)) return 1;
    if (check_host_index()) return 1;
    printf("host_sanitize: clean\n");
    return 0;
}
