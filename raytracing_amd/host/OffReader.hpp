/*
 * OffReader.hpp -- a fast reader for the plain "OFF" files Mesh::loadOFF takes (R/src/Mesh.cpp:21-39).
 *
 * The reference reads the file token by token with operator>>: tag, three counts, 3 * nv floats, then per
 * face one count (ignored) and three indices.  That costs ~1 us per triangle; this reader maps the whole file,
 * cuts it into one piece per thread at white space, counts the tokens of each piece, and lets every thread
 * convert its own tokens with std::from_chars, addressing the output by global token number.
 *
 * It only ever returns what the token-by-token reader would have produced: operator>> converts a float with
 * strtof and std::from_chars is correctly rounded as well, so a plain decimal token gives the same bits.
 * Anything else -- another tag (COFF ...), a token with a character outside [0-9.eE+-], a leading '+', a
 * conversion that does not consume its token or reports a range error, too few tokens -- makes read() return
 * false without touching the output, and the caller runs the reference's reader instead.
 */
#pragma once
#include <atomic>
#include <charconv>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace offreader {

inline bool is_space(char c) { return c == ' ' || c == '\n' || c == '\t' || c == '\r' || c == '\v' || c == '\f'; }

inline bool plain_number(const char* b, const char* e, bool allow_float) {
    if (b == e) return false;
    for (const char* p = b; p < e; ++p) {
        const char c = *p;
        bool ok = (c >= '0' && c <= '9') || c == '-' || (allow_float && (c == '.' || c == 'e' || c == 'E'));
        if (c == '+') ok = allow_float && p > b && (p[-1] == 'e' || p[-1] == 'E'); /* only as the exponent's sign */
        if (!ok) return false;
    }
    return true;
}

/* Short decimals without the library: a token [-]digits[.digits][e[-]digits] whose digit string (leading zeros
 * aside) is an integer m < 2^24 and whose decimal exponent k has |k| <= 10 is m * 10^k or m / 10^-k with m and
 * the power of ten both exact in binary32, so the one IEEE operation rounds the true value correctly -- the
 * result strtof returns.  Anything longer is left to std::from_chars. */
inline bool short_decimal(const char* b, const char* e, float& out) {
    static const float p10[11] = {1e0f, 1e1f, 1e2f, 1e3f, 1e4f, 1e5f, 1e6f, 1e7f, 1e8f, 1e9f, 1e10f};
    const char* p = b;
    bool neg = false;
    if (p < e && *p == '-') { neg = true; ++p; }
    unsigned int m = 0;
    int digits = 0, frac = 0;
    bool dot = false, any = false;
    for (; p < e; ++p) {
        const char c = *p;
        if (c >= '0' && c <= '9') {
            any = true;
            if (m == 0 && c == '0') { if (dot) ++frac; continue; } /* leading zeros carry no digits */
            if (++digits > 8) return false;
            m = m * 10 + (unsigned)(c - '0');
            if (dot) ++frac;
        } else if (c == '.' && !dot) dot = true;
        else break;
    }
    if (!any || m >= (1u << 24)) return false;
    int ex = 0;
    if (p < e) {
        if (*p != 'e' && *p != 'E') return false;
        ++p;
        bool eneg = false;
        if (p < e && (*p == '-' || *p == '+')) { eneg = *p == '-'; ++p; }
        if (p == e || e - p > 3) return false;
        for (; p < e; ++p) {
            if (*p < '0' || *p > '9') return false;
            ex = ex * 10 + (*p - '0');
        }
        if (eneg) ex = -ex;
    }
    const int k = ex - frac;
    if (k < -10 || k > 10) return false;
    const float v = k < 0 ? (float)m / p10[-k] : (float)m * p10[k];
    if (v > 3.4e38f) return false;
    out = neg ? -v : v;
    return true;
}

inline bool to_float(const char* b, const char* e, float& out) {
    if (!plain_number(b, e, true)) return false;
    if (short_decimal(b, e, out)) return true;
    float v;
    auto r = std::from_chars(b, e, v);
    if (r.ec != std::errc() || r.ptr != e) return false;
    out = v;
    return true;
}

inline bool to_uint(const char* b, const char* e, unsigned int& out) { /* operator>>(unsigned) accepts "-1" and wraps: not here */
    if (!plain_number(b, e, false) || *b == '-') return false;
    unsigned int v;
    auto r = std::from_chars(b, e, v);
    if (r.ec != std::errc() || r.ptr != e) return false;
    out = v;
    return true;
}

inline bool to_int(const char* b, const char* e, int& out) {
    if (!plain_number(b, e, false)) return false;
    int v;
    auto r = std::from_chars(b, e, v);
    if (r.ec != std::errc() || r.ptr != e) return false;
    out = v;
    return true;
}

/* positions: 3 * nv floats, indices: 3 * nt, in file order.  false = not handled, outputs unspecified. */
inline bool read(const std::string& filename, std::vector<float>& positions, std::vector<unsigned int>& indices) {
    std::FILE* f = std::fopen(filename.c_str(), "rb");
    if (!f) return false;
    std::string buf;
    if (std::fseek(f, 0, SEEK_END) == 0) {
        long n = std::ftell(f);
        if (n > 0) buf.resize((size_t)n);
        std::rewind(f);
    }
    const size_t got = buf.empty() ? 0 : std::fread(&buf[0], 1, buf.size(), f);
    std::fclose(f);
    if (got != buf.size() || buf.empty()) return false;
    const char* p = buf.data();
    const char* end = p + buf.size();
    auto next_token = [&](const char*& b, const char*& e) {
        while (p < end && is_space(*p)) ++p;
        b = p;
        while (p < end && !is_space(*p)) ++p;
        e = p;
        return b < e;
    };
    const char *b, *e;
    if (!next_token(b, e) || e - b != 3 || std::memcmp(b, "OFF", 3) != 0) return false;
    unsigned int nv, nt, ne;
    if (!next_token(b, e) || !to_uint(b, e, nv)) return false;
    if (!next_token(b, e) || !to_uint(b, e, nt)) return false;
    if (!next_token(b, e) || !to_uint(b, e, ne)) return false;
    const size_t need = 3 * (size_t)nv + 4 * (size_t)nt;
    const char* body = p;
    const size_t body_len = (size_t)(end - body);

    unsigned n_threads = std::thread::hardware_concurrency();
    n_threads = n_threads == 0 ? 1 : (n_threads > 16 ? 16 : n_threads);
    if (body_len < (1u << 20)) n_threads = 1;
    /* piece k = [cut[k], cut[k+1]), cut at white space so no token straddles two pieces */
    std::vector<const char*> cut(n_threads + 1);
    cut[0] = body; cut[n_threads] = end;
    for (unsigned k = 1; k < n_threads; ++k) {
        const char* c = body + body_len / n_threads * k;
        if (c < cut[k - 1]) c = cut[k - 1];
        while (c < end && !is_space(*c)) ++c;
        cut[k] = c;
    }
    std::vector<size_t> count(n_threads, 0);
    auto run = [&](auto&& fn) {
        std::vector<std::thread> pool;
        for (unsigned k = 1; k < n_threads; ++k) pool.emplace_back(fn, k);
        fn(0u);
        for (std::thread& t : pool) t.join();
    };
    run([&](unsigned k) {
        size_t n = 0;
        bool in_token = false;
        for (const char* c = cut[k]; c < cut[k + 1]; ++c) {
            const bool sp = is_space(*c);
            n += (!sp && !in_token) ? 1 : 0;
            in_token = !sp;
        }
        count[k] = n;
    });
    std::vector<size_t> first(n_threads + 1, 0);
    for (unsigned k = 0; k < n_threads; ++k) first[k + 1] = first[k] + count[k];
    if (first[n_threads] < need) return false; /* the token-by-token reader would run into the end of the file */

    positions.assign(3 * (size_t)nv, 0.f);
    indices.assign(3 * (size_t)nt, 0u);
    std::atomic<bool> bad{false};
    run([&](unsigned k) {
        size_t g = first[k];
        const char* c = cut[k];
        const char* stop = cut[k + 1];
        while (g < need && !bad.load(std::memory_order_relaxed)) {
            while (c < stop && is_space(*c)) ++c;
            if (c >= stop) break;
            const char* tb = c;
            while (c < stop && !is_space(*c)) ++c;
            bool ok;
            if (g < 3 * (size_t)nv) ok = to_float(tb, c, positions[g]);
            else {
                const size_t q = g - 3 * (size_t)nv;
                int corners;
                if (q % 4 == 0) ok = to_int(tb, c, corners); /* read and ignored, as in the reference */
                else ok = to_uint(tb, c, indices[3 * (q / 4) + (q % 4 - 1)]);
            }
            if (!ok) bad.store(true, std::memory_order_relaxed);
            ++g;
        }
    });
    return !bad.load();
}

}  // namespace offreader
