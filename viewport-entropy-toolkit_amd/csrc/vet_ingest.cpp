// Host-side track loader of the C-ABI (include/vet.h, vet_csv_read_tracks): the three columns the
// reference reads from every user's CSV (`pd.read_csv(...)[time, 2dmu, 2dmv]`,
// utilities/data_utils.py:305-316) parsed on all host cores, so that ingest keeps up with the GPU.
//
// The decimal -> FP64 conversion follows the published algorithm of pandas' default C-engine
// converter ("high" precision: up to 17 significant digits accumulated in a double, then ONE
// multiply or divide by a power of ten), because the pixel quantiser truncates `mu * W` and a
// last-bit difference can move a sample to another pixel.  Anything outside plain unquoted numeric
// CSV (quotes, text, inf, odd row shapes, > 17-digit integers, ...) is reported as
// VET_CSV_FALLBACK and the caller parses that file with pandas itself; the Python side also
// cross-checks the first file a process reads against pandas.
//
// Host code only: no HIP calls, usable without a GPU.

#include "../../include/vet.h"

#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>
#include <thread>
#include <vector>

namespace {

const double kPow10[] = {
    1e0,   1e1,   1e2,   1e3,   1e4,   1e5,   1e6,   1e7,   1e8,   1e9,   1e10,  1e11,  1e12,  1e13,  1e14,  1e15,
    1e16,  1e17,  1e18,  1e19,  1e20,  1e21,  1e22,  1e23,  1e24,  1e25,  1e26,  1e27,  1e28,  1e29,  1e30,  1e31,
    1e32,  1e33,  1e34,  1e35,  1e36,  1e37,  1e38,  1e39,  1e40,  1e41,  1e42,  1e43,  1e44,  1e45,  1e46,  1e47,
    1e48,  1e49,  1e50,  1e51,  1e52,  1e53,  1e54,  1e55,  1e56,  1e57,  1e58,  1e59,  1e60,  1e61,  1e62,  1e63,
    1e64,  1e65,  1e66,  1e67,  1e68,  1e69,  1e70,  1e71,  1e72,  1e73,  1e74,  1e75,  1e76,  1e77,  1e78,  1e79,
    1e80,  1e81,  1e82,  1e83,  1e84,  1e85,  1e86,  1e87,  1e88,  1e89,  1e90,  1e91,  1e92,  1e93,  1e94,  1e95,
    1e96,  1e97,  1e98,  1e99,  1e100};
constexpr int kMaxPow = 100;          // larger decimal exponents go to the pandas fallback
constexpr int kMaxDigits = 17;

inline bool is_digit(char c) { return c >= '0' && c <= '9'; }

// One numeric field [p, end).  Returns 0 = number, 1 = missing value, 2 = not plain numeric.
int parse_field(const char* p, const char* end, double* out) {
    const size_t len = (size_t)(end - p);
    if (len == 0) { *out = std::numeric_limits<double>::quiet_NaN(); return 1; }
    // pandas' default missing-value spellings
    static const char* const kNA[] = {"NaN", "nan", "NA", "N/A", "n/a", "NULL", "null", "None", "<NA>", "#N/A",
                                      "#NA", "-NaN", "-nan", "#N/A N/A", "1.#IND", "-1.#IND", "1.#QNAN", "-1.#QNAN"};
    if (!is_digit(p[len - 1]) && p[len - 1] != '.') {
        for (const char* na : kNA)
            if (strlen(na) == len && memcmp(na, p, len) == 0) { *out = std::numeric_limits<double>::quiet_NaN(); return 1; }
        return 2;
    }
    bool negative = false;
    if (*p == '-') { negative = true; ++p; } else if (*p == '+') { ++p; }
    double number = 0.0;
    int exponent = 0, num_digits = 0, num_decimals = 0;
    bool any_digit = false, has_point = false, has_exp = false;
    // pandas tries a column as int64, then uint64, before float64, and an integer part that overflows both
    // makes it give up and keep the column as text (observed with pandas 2.3: 18446744073709551615e0 is a
    // float, 18446744073709551616e0 and -9223372036854775809.5 are not; whether it happens depends on the
    // rows before): leave every such file to pandas
    const char* int_begin = p;
    while (int_begin < end && *int_begin == '0') ++int_begin;
    while (p < end && is_digit(*p)) {
        any_digit = true;
        if (num_digits < kMaxDigits) { number = number * 10.0 + (double)(*p - '0'); ++num_digits; }
        else ++exponent;
        ++p;
    }
    {
        static const char kU64Max[] = "18446744073709551615", kI64Min[] = "9223372036854775808";
        const size_t int_len = p > int_begin ? (size_t)(p - int_begin) : 0;
        const char* lim = negative ? kI64Min : kU64Max;
        const size_t lim_len = negative ? 19 : 20;
        if (int_len > lim_len || (int_len == lim_len && memcmp(int_begin, lim, lim_len) > 0)) return 2;
    }
    if (p < end && *p == '.') {
        has_point = true;
        ++p;
        while (p < end && num_digits < kMaxDigits && is_digit(*p)) {
            any_digit = true;
            number = number * 10.0 + (double)(*p - '0');
            ++p; ++num_digits; ++num_decimals;
        }
        while (p < end && is_digit(*p)) { any_digit = true; ++p; }      // digits beyond the 17th are dropped
        exponent -= num_decimals;
    }
    if (!any_digit) return 2;
    if (negative) number = -number;
    if (p < end && (*p == 'e' || *p == 'E')) {
        has_exp = true;
        ++p;
        bool eneg = false;
        if (p < end && (*p == '-' || *p == '+')) { eneg = *p == '-'; ++p; }
        if (p >= end || !is_digit(*p)) return 2;
        int n = 0;
        while (p < end && is_digit(*p)) { if (n < 100000) n = n * 10 + (*p - '0'); ++p; }
        exponent += eneg ? -n : n;
    }
    if (p != end) return 2;
    // integers of more than 15 digits are int64 columns in pandas (exact), not this converter's job
    if (!has_point && !has_exp && num_digits + (exponent > 0 ? exponent : 0) > 15) return 2;
    if (exponent > kMaxPow || exponent < -kMaxPow) return 2;
    if (exponent > 0) number *= kPow10[exponent];
    else if (exponent < 0) number /= kPow10[-exponent];
    *out = number;
    return 0;
}

struct Columns { std::vector<double> t, a, b; };

// returns VET_CSV_OK / VET_CSV_FALLBACK / VET_CSV_IO
int parse_file(const char* path, Columns& col, std::string& buf) {
    FILE* f = fopen(path, "rb");
    if (!f) return VET_CSV_IO;
    buf.clear();                                     // capacity is kept from file to file
    {
        char chunk[1 << 16];
        size_t n;
        while ((n = fread(chunk, 1, sizeof chunk, f)) > 0) buf.append(chunk, n);
    }
    const bool io_error = ferror(f) != 0;
    fclose(f);
    if (io_error) return VET_CSV_IO;
    col.t.clear(); col.a.clear(); col.b.clear();
    if (buf.find('"') != std::string::npos || buf.find('\0') != std::string::npos) return VET_CSV_FALLBACK;
    const char* p = buf.data();
    const char* const end = p + buf.size();
    if (end - p >= 3 && (unsigned char)p[0] == 0xEF && (unsigned char)p[1] == 0xBB && (unsigned char)p[2] == 0xBF) p += 3;

    auto line_end = [&](const char* s) { while (s < end && *s != '\n' && *s != '\r') ++s; return s; };
    auto next_line = [&](const char* e) {            // e points at '\n', '\r' or end
        if (e < end && *e == '\r') { ++e; if (e < end && *e == '\n') ++e; return e; }
        if (e < end) ++e;
        return e;
    };
    // header: first non-blank line
    const char* le = line_end(p);
    while (p < end && le == p) { p = next_line(le); le = line_end(p); }
    if (p >= end) return VET_CSV_FALLBACK;
    int idx[3] = {-1, -1, -1};
    static const char* const names[3] = {"time", "2dmu", "2dmv"};
    int n_header = 0;
    for (const char* s = p;; ) {
        const char* e = s;
        while (e < le && *e != ',') ++e;
        for (int k = 0; k < 3; ++k)
            if ((size_t)(e - s) == strlen(names[k]) && memcmp(s, names[k], (size_t)(e - s)) == 0) {
                if (idx[k] >= 0) return VET_CSV_FALLBACK;          // duplicate column name
                idx[k] = n_header;
            }
        ++n_header;
        if (e >= le) break;
        s = e + 1;
    }
    if (idx[0] < 0 || idx[1] < 0 || idx[2] < 0) return VET_CSV_FALLBACK;   // pandas words the error
    p = next_line(le);
    const size_t guess = (size_t)(end - p) / 24 + 16;
    col.t.reserve(guess); col.a.reserve(guess); col.b.reserve(guess);
    const double nan = std::numeric_limits<double>::quiet_NaN();
    while (p < end) {
        le = line_end(p);
        if (le == p) { p = next_line(le); continue; }               // blank lines are skipped
        double v[3] = {nan, nan, nan};
        int field = 0;
        bool only_space = true;
        for (const char* s = p;; ) {
            const char* e = s;
            while (e < le && *e != ',') ++e;
            for (const char* c = s; c < e; ++c) if (*c != ' ' && *c != '\t') only_space = false;
            for (int k = 0; k < 3; ++k)
                if (field == idx[k] && parse_field(s, e, &v[k]) == 2) return VET_CSV_FALLBACK;
            ++field;
            if (e >= le) break;
            s = e + 1;
        }
        if (field > n_header) return VET_CSV_FALLBACK;              // pandas would infer an index column
        if (field == 1 && only_space) return VET_CSV_FALLBACK;      // whitespace-only line: leave to pandas
        col.t.push_back(v[0]); col.a.push_back(v[1]); col.b.push_back(v[2]);
        p = next_line(le);
    }
    return VET_CSV_OK;
}

double* steal(const std::vector<double>& v) {
    double* out = (double*)malloc((v.size() ? v.size() : 1) * sizeof(double));
    if (out && !v.empty()) memcpy(out, v.data(), v.size() * sizeof(double));
    return out;
}

}  // namespace

extern "C" {

int vet_csv_read_tracks(int n_files, const char* const* paths, vet_track* tracks, int n_threads) {
    if (n_files < 0 || (n_files && (!paths || !tracks))) return VET_ERR_INVALID;
    for (int i = 0; i < n_files; ++i) tracks[i] = vet_track{nullptr, nullptr, nullptr, 0, VET_CSV_IO};
    std::atomic<int> next{0};
    auto work = [&]() {
        Columns col;                                 // per-thread scratch, reused across files
        std::string buf;
        for (int i; (i = next.fetch_add(1)) < n_files;) {
            int st = paths[i] ? parse_file(paths[i], col, buf) : VET_CSV_IO;
            if (st == VET_CSV_OK) {
                tracks[i].time = steal(col.t); tracks[i].mu = steal(col.a); tracks[i].mv = steal(col.b);
                tracks[i].n_rows = (int64_t)col.t.size();
                if (!tracks[i].time || !tracks[i].mu || !tracks[i].mv) st = VET_CSV_IO;
            }
            tracks[i].status = st;
        }
    };
    int nt = n_threads > 0 ? n_threads : (int)std::thread::hardware_concurrency();
    if (nt < 1) nt = 1;
    if (nt > n_files) nt = n_files;
    if (nt > 64) nt = 64;
    std::vector<std::thread> pool;
    for (int t = 1; t < nt; ++t) pool.emplace_back(work);
    work();
    for (auto& th : pool) th.join();
    return VET_OK;
}

void vet_csv_free_tracks(int n_files, vet_track* tracks) {
    if (!tracks) return;
    for (int i = 0; i < n_files; ++i) {
        free(tracks[i].time); free(tracks[i].mu); free(tracks[i].mv);
        tracks[i].time = tracks[i].mu = tracks[i].mv = nullptr;
        tracks[i].n_rows = 0;
    }
}

}  // extern "C"
