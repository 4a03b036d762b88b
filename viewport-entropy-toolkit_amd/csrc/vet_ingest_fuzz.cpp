// Sanitizer / fuzz driver of the host-side track loader (vet_ingest.cpp, include/vet.h: vet_csv_read_tracks).
// Built by `make asan` with -fsanitize=address,undefined (host code only: GPU sanitizers are not available)
// into lib/vet_ingest_asan; tests/test_csv_loader_asan.py feeds it the regression corpus and hypothesis-made
// byte soup and compares every line with the production library's answer.
//
//   vet_ingest_asan FILE...   ->  one line per file:  <status> <n_rows> <fnv1a-64 of time|mu|mv bytes>
// The files are parsed twice — one call per file on one thread, then one call for all files on four
// threads — and the two answers must agree (exit code 3 otherwise; a sanitizer report aborts non-zero).
#include "../../include/vet.h"

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

static uint64_t fnv(uint64_t h, const void* p, size_t n) {
    const unsigned char* c = (const unsigned char*)p;
    for (size_t i = 0; i < n; ++i) { h ^= c[i]; h *= 0x100000001B3ull; }
    return h;
}

static uint64_t digest(const vet_track& t) {
    uint64_t h = 0xCBF29CE484222325ull;
    if (t.status != VET_CSV_OK) return h;
    const size_t b = (size_t)t.n_rows * sizeof(double);
    h = fnv(h, t.time, b); h = fnv(h, t.mu, b); h = fnv(h, t.mv, b);
    return h;
}

int main(int argc, char** argv) {
    const int n = argc - 1;
    if (n <= 0) return 0;
    std::vector<vet_track> all((size_t)n), one(1);
    if (vet_csv_read_tracks(n, argv + 1, all.data(), 4) != VET_OK) return 2;
    int rc = 0;
    for (int i = 0; i < n; ++i) {
        if (vet_csv_read_tracks(1, argv + 1 + i, one.data(), 1) != VET_OK) return 2;
        const uint64_t a = digest(all[(size_t)i]), b = digest(one[0]);
        if (a != b || all[(size_t)i].status != one[0].status || all[(size_t)i].n_rows != one[0].n_rows) rc = 3;
        printf("%d %lld %016llx\n", one[0].status, (long long)one[0].n_rows, (unsigned long long)b);
        vet_csv_free_tracks(1, one.data());
    }
    vet_csv_free_tracks(n, all.data());
    // NULL / empty argument handling
    if (vet_csv_read_tracks(0, nullptr, nullptr, 0) != VET_OK) rc = 4;
    if (vet_csv_read_tracks(1, nullptr, nullptr, 0) != VET_ERR_INVALID) rc = 4;
    vet_csv_free_tracks(0, nullptr);
    return rc;
}
