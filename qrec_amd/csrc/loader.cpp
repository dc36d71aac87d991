// Native reader of the reference's on-disk rating format (util/io.py:31-76): one "user item rating" record per
// line, fields split on EVERY single delimiter character (the reference's regex ' |,|\t' -- two spaces in a
// row make an empty field, exactly as re.split does), `-columns` picks the fields, `-header` drops the first
// line, `-b t` drops records rated below t and sets the rest to 1.  Output: dense ids in first-appearance order
// plus the name tables, i.e. the arrays the kernels want, without 1.2 M three-element Python lists (2.4 s +
// 2.0 s of host time at the Yelp2018 shape before a single kernel runs).
//
// Contract with the host mirror (qrec_amd/util/io.py): anything this parser cannot reproduce EXACTLY as
// CPython would -- non-ASCII bytes (str.strip / decoding rules), a rating token that is not a plain decimal
// literal, a record with too few fields (the reference raises there) -- returns QREC_ERR_UNSUPPORTED and the
// caller takes the pure-Python path, which then behaves (and fails) as the reference does.
#include <cerrno>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <string_view>
#include <vector>

#include "common.h"

using namespace qrec;

struct qrec_ratings {
    std::vector<int32_t> user, item;
    std::vector<double> rating;
    std::string names[2];          // '\n'-joined, first-appearance order
    int32_t count[2] = {0, 0};
};

namespace {

inline bool is_space(unsigned char c) { return c == ' ' || (c >= 9 && c <= 13) || (c >= 28 && c <= 31); }  // str.strip(), ASCII

// [+-]? (digits [. digits*] | . digits) ([eE] [+-]? digits)?   -- the literals float() and strtod agree on
bool plain_decimal(std::string_view t) {
    size_t k = 0, n = t.size();
    if (k < n && (t[k] == '+' || t[k] == '-')) k++;
    size_t d0 = k;
    while (k < n && t[k] >= '0' && t[k] <= '9') k++;
    size_t int_digits = k - d0, frac_digits = 0;
    if (k < n && t[k] == '.') {
        k++;
        size_t f0 = k;
        while (k < n && t[k] >= '0' && t[k] <= '9') k++;
        frac_digits = k - f0;
    }
    if (int_digits + frac_digits == 0) return false;
    if (k < n && (t[k] == 'e' || t[k] == 'E')) {
        k++;
        if (k < n && (t[k] == '+' || t[k] == '-')) k++;
        size_t e0 = k;
        while (k < n && t[k] >= '0' && t[k] <= '9') k++;
        if (k == e0) return false;
    }
    return k == n;
}

// name -> dense id in first-appearance order.  Open addressing, 16-byte slots that hold the name's first eight bytes: the
// usual name ("u1234", an integer id) is compared without touching the file buffer -- std::unordered_map<string_view>
// compared every probe against the bytes where the name first occurred, a cache miss apiece: 110-130 ns per lookup, 270 of
// the loader's 400 ms at the Yelp2018 shape.  Files are usually grouped by user: the previous row's name is tried first.
struct Interner {
    struct Slot { uint64_t head; uint32_t len; int32_t id; };
    std::vector<Slot> tab;
    size_t mask = 0;
    std::vector<std::string_view> order;
    std::string_view last;
    int32_t last_id = -1;

    static uint64_t head8(std::string_view s) {
        uint64_t h = 0;
        std::memcpy(&h, s.data(), s.size() < 8 ? s.size() : 8);
        return h;
    }
    static uint64_t hash(uint64_t head, std::string_view s) {
        uint64_t h = (head ^ (uint64_t)s.size() * 0x9E3779B97F4A7C15ull) * 0xD6E8FEB86659FD93ull;
        for (size_t k = 8; k < s.size(); k++) { h ^= (unsigned char)s[k]; h *= 1099511628211ull; }
        return h ^ (h >> 32);
    }
    void reserve(size_t n_names) {
        size_t cap = 1024;
        while (cap < 2 * n_names) cap <<= 1;
        tab.assign(cap, Slot{0, 0, -1});
        mask = cap - 1;
    }
    bool same(const Slot &sl, uint64_t head, std::string_view s) const {
        if (sl.head != head || sl.len != (uint32_t)s.size()) return false;
        return s.size() <= 8 || std::memcmp(order[sl.id].data() + 8, s.data() + 8, s.size() - 8) == 0;
    }
    void grow() {
        std::vector<Slot> old;
        old.swap(tab);
        tab.assign(old.size() * 2, Slot{0, 0, -1});
        mask = tab.size() - 1;
        for (const Slot &sl : old)
            if (sl.id >= 0) {
                size_t k = hash(sl.head, order[sl.id]) & mask;
                while (tab[k].id >= 0) k = (k + 1) & mask;
                tab[k] = sl;
            }
    }
    int32_t id(std::string_view s) {
        if (last_id >= 0 && s.size() == last.size() && std::memcmp(s.data(), last.data(), s.size()) == 0) return last_id;
        if (tab.empty()) reserve(512);
        const uint64_t head = head8(s);
        size_t k = hash(head, s) & mask;
        while (tab[k].id >= 0) {
            if (same(tab[k], head, s)) { last = order[tab[k].id]; return last_id = tab[k].id; }
            k = (k + 1) & mask;
        }
        const int32_t v = (int32_t)order.size();
        order.push_back(s);
        tab[k] = Slot{head, (uint32_t)s.size(), v};
        if (order.size() * 2 > tab.size()) grow();
        last = s;
        return last_id = v;
    }
    void join(std::string &out) const {
        size_t bytes = 0;
        for (auto s : order) bytes += s.size() + 1;
        out.clear();
        out.reserve(bytes);
        for (size_t k = 0; k < order.size(); k++) {
            if (k) out.push_back('\n');
            out.append(order[k]);
        }
    }
};

}  // namespace

extern "C" {

int qrec_ratings_load(const char *path, const char *delims, int32_t col_user, int32_t col_item, int32_t col_rating,
                      int32_t skip_header, int32_t binarize, double threshold, qrec_ratings **out) {
    QREC_REQUIRE(path && out && col_user >= 0 && col_item >= 0, "qrec_ratings_load: bad arguments");
    *out = nullptr;
    if (binarize && col_rating < 0) { set_error("qrec_ratings_load: -b needs a rating column"); return QREC_ERR_UNSUPPORTED; }
    const char *dl = (delims && *delims) ? delims : " ,\t";
    bool is_delim[256] = {false};
    for (const unsigned char *p = (const unsigned char *)dl; *p; p++) {
        if (*p >= 0x80 || *p == '\n' || *p == '\r') { set_error("qrec_ratings_load: unsupported delimiter"); return QREC_ERR_UNSUPPORTED; }
        is_delim[*p] = true;
    }
    FILE *f = std::fopen(path, "rb");
    if (!f) { set_error("qrec_ratings_load: cannot open %s: %s", path, std::strerror(errno)); return QREC_ERR_INVALID; }
    std::string buf;
    {
        std::fseek(f, 0, SEEK_END);
        const long sz = std::ftell(f);
        std::fseek(f, 0, SEEK_SET);
        buf.resize(sz > 0 ? (size_t)sz : 0);
        const size_t got = buf.empty() ? 0 : std::fread(&buf[0], 1, buf.size(), f);
        std::fclose(f);
        if (got != buf.size()) { set_error("qrec_ratings_load: short read on %s", path); return QREC_ERR_INVALID; }
    }
    for (unsigned char c : buf)
        if (c >= 0x80 || c == 0) { set_error("qrec_ratings_load: non-ASCII byte in %s", path); return QREC_ERR_UNSUPPORTED; }

    auto *res = new qrec_ratings();
    Interner users, items;
    {
        size_t lines = 1;
        for (char c : buf) lines += (c == '\n');
        res->user.reserve(lines); res->item.reserve(lines); res->rating.reserve(lines);
        users.reserve(lines / 32 + 64); items.reserve(lines / 32 + 64);          // a guess; the tables double as they fill
    }
    const int need = (col_user > col_item ? col_user : col_item) > col_rating ? (col_user > col_item ? col_user : col_item) : col_rating;
    std::vector<std::string_view> fields;
    const char *p = buf.data(), *end = p + buf.size();
    int64_t line_no = 0;
    while (p < end) {
        // universal newlines, as text-mode readlines(): \n, \r\n, \r
        const char *q = p;
        while (q < end && *q != '\n' && *q != '\r') q++;
        const char *next = q;
        if (next < end) next += (*q == '\r' && q + 1 < end && q[1] == '\n') ? 2 : 1;
        const char *a = p, *b = q;
        p = next;
        if (line_no++ == 0 && skip_header) continue;
        while (a < b && is_space((unsigned char)*a)) a++;
        while (b > a && is_space((unsigned char)b[-1])) b--;
        fields.clear();
        const char *s = a;
        for (const char *c = a; c <= b; c++) {
            if (c == b || is_delim[(unsigned char)*c]) { fields.emplace_back(s, (size_t)(c - s)); s = c + 1; }
        }
        if ((int)fields.size() <= need) {   // the reference raises IndexError here
            set_error("qrec_ratings_load: line %lld of %s has %d fields", (long long)line_no, path, (int)fields.size());
            delete res;
            return QREC_ERR_UNSUPPORTED;
        }
        double r = 1.0;
        if (col_rating >= 0) {
            std::string_view t = fields[col_rating];
            while (!t.empty() && is_space((unsigned char)t.front())) t.remove_prefix(1);
            while (!t.empty() && is_space((unsigned char)t.back())) t.remove_suffix(1);
            if (!plain_decimal(t)) {
                set_error("qrec_ratings_load: line %lld of %s: rating is not a plain decimal literal", (long long)line_no, path);
                delete res;
                return QREC_ERR_UNSUPPORTED;
            }
            if (t.size() == 1) {
                r = (double)(t[0] - '0');           // the common case: "1" .. "5"
            } else {
                char tmp[64];
                if (t.size() >= sizeof(tmp)) { set_error("qrec_ratings_load: rating literal too long"); delete res; return QREC_ERR_UNSUPPORTED; }
                std::memcpy(tmp, t.data(), t.size());
                tmp[t.size()] = 0;
                r = std::strtod(tmp, nullptr);      // correctly rounded, like float()
            }
        }
        if (binarize) {
            if (r < threshold) continue;
            r = 1.0;
        }
        res->user.push_back(users.id(fields[col_user]));
        res->item.push_back(items.id(fields[col_item]));
        res->rating.push_back(r);
    }
    users.join(res->names[0]);
    items.join(res->names[1]);
    res->count[0] = (int32_t)users.order.size();
    res->count[1] = (int32_t)items.order.size();
    *out = res;
    return QREC_OK;
}

int64_t qrec_ratings_rows(const qrec_ratings *h) { return h ? (int64_t)h->user.size() : -1; }
int32_t qrec_ratings_count(const qrec_ratings *h, int32_t which) { return (h && (which == 0 || which == 1)) ? h->count[which] : -1; }
int64_t qrec_ratings_names_bytes(const qrec_ratings *h, int32_t which) {
    return (h && (which == 0 || which == 1)) ? (int64_t)h->names[which].size() : -1;
}
int qrec_ratings_copy(const qrec_ratings *h, int32_t *user_out, int32_t *item_out, double *rating_out) {
    QREC_REQUIRE(h && user_out && item_out && rating_out, "qrec_ratings_copy: null argument");
    std::memcpy(user_out, h->user.data(), h->user.size() * sizeof(int32_t));
    std::memcpy(item_out, h->item.data(), h->item.size() * sizeof(int32_t));
    std::memcpy(rating_out, h->rating.data(), h->rating.size() * sizeof(double));
    return QREC_OK;
}
int qrec_ratings_names(const qrec_ratings *h, int32_t which, char *out) {
    QREC_REQUIRE(h && out && (which == 0 || which == 1), "qrec_ratings_names: bad arguments");
    std::memcpy(out, h->names[which].data(), h->names[which].size());
    return QREC_OK;
}
void qrec_ratings_free(qrec_ratings *h) { delete h; }

}  // extern "C"
