// Host-side text formatting of `.index.json` (include/lcrec.h: lcrec_index_json_*).
//
// The reference builds one Python string per token and one list per item, then json.dump's a dict of
// N lists (index/generate_indices.py:83-92,138-145): minutes of interpreter time at 10 M items.
// Here a chunk of the int64 [n][L] index matrix becomes its JSON text in one pass over memory.
#include "common.h"

#include <thread>
#include <vector>

namespace lcrec {

// decimal digits of a signed 64-bit value, written forwards; returns the advanced pointer
static inline char *put_i64(char *p, int64_t v)
{
    uint64_t u = (uint64_t)v;
    if (v < 0) {
        *p++ = '-';
        u = 0 - u;
    }
    char tmp[20];
    int d = 0;
    do {
        tmp[d++] = (char)('0' + u % 10);
        u /= 10;
    } while (u);
    while (d) *p++ = tmp[--d];
    return p;
}

// "<id>": ["<a_i>", "<b_j>", ...]   -- at most 24 + L * 28 bytes
static inline char *put_item(char *p, int64_t item, const int64_t *row, int L)
{
    *p++ = '"';
    p = put_i64(p, item);
    *p++ = '"'; *p++ = ':'; *p++ = ' '; *p++ = '[';
    for (int l = 0; l < L; ++l) {
        if (l) { *p++ = ','; *p++ = ' '; }
        *p++ = '"'; *p++ = '<'; *p++ = (char)('a' + l); *p++ = '_';
        p = put_i64(p, row[l]);
        *p++ = '>'; *p++ = '"';
    }
    *p++ = ']';
    return p;
}

constexpr int64_t ITEM_HEAD = 26;   // quote + 20 digits (incl. sign) + quote, colon, space, bracket ... + "]"
constexpr int64_t TOKEN_MAX = 28;   // `, "<x_` (7) + 20 digits + `>"` (2), rounded

int64_t index_json_bound(int64_t n, int L) { return n * (ITEM_HEAD + (int64_t)L * TOKEN_MAX + 2) + 2; }

int64_t index_json_format(const int64_t *idx, int64_t n, int L, int64_t first_item, char *out, int64_t cap)
{
    if (n == 0) return 0;
    if (!idx || !out) return fail(LCREC_EINVAL, "index_json_format: NULL pointer");
    if (n < 0 || L < 1 || L > 26) return fail(LCREC_EINVAL, "index_json_format: need n >= 0 and 1 <= L <= 26 (got L=%d)", L);
    // Fast path: the caller's buffer covers the worst case, so the rows can be cut into independent
    // slices, each formatted by its own thread into its own worst-case window, then compacted.
    const int64_t per_item = ITEM_HEAD + (int64_t)L * TOKEN_MAX + 2;
    if (cap >= n * per_item) {
        unsigned hw = std::thread::hardware_concurrency();
        int nt = (int)(hw ? (hw > 16 ? 16 : hw) : 1);
        if (n < 65536) nt = 1;
        std::vector<int64_t> len(nt, 0);
        auto work = [&](int t) {
            const int64_t lo = n * t / nt, hi = n * (t + 1) / nt;
            char *p = out + lo * per_item;
            for (int64_t i = lo; i < hi; ++i) {
                if (i > lo) { *p++ = ','; *p++ = ' '; }
                p = put_item(p, first_item + i, idx + i * L, L);
            }
            len[t] = p - (out + lo * per_item);
        };
        if (nt == 1) {
            work(0);
            return len[0];
        }
        std::vector<std::thread> th;
        for (int t = 0; t < nt; ++t) th.emplace_back(work, t);
        for (auto &x : th) x.join();
        char *w = out;
        for (int t = 0; t < nt; ++t) {
            const int64_t lo = n * t / nt;
            if (len[t] == 0) continue;
            if (w != out) { *w++ = ','; *w++ = ' '; }
            memmove(w, out + lo * per_item, (size_t)len[t]);   // destinations never overtake sources
            w += len[t];
        }
        return w - out;
    }
    // Tight buffer: single pass with a bound check per item.
    char *p = out, *end = out + cap;
    for (int64_t i = 0; i < n; ++i) {
        if (end - p < per_item) return fail(LCREC_EWORKSPACE, "index_json_format: buffer of %lld bytes is too small", (long long)cap);
        if (i) { *p++ = ','; *p++ = ' '; }
        p = put_item(p, first_item + i, idx + i * L, L);
    }
    return p - out;
}

}  // namespace lcrec
