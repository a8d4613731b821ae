// ORACLE (test infrastructure, NOT product code) -- CPU restatement of the reference's FFT,
// polynomial and quartic helpers. Only tests/, smoke() and bench.py's cpu_baseline may use it.
//
// Follows /root/reference/src/math/fft.rs (fft_in_place :16, get_twiddles :58, get_inv_twiddles :66,
// permute :71, butterflies :92-108), src/math/polynom.rs (eval :9, eval_fft_twiddles :34,
// interpolate :47, interpolate_fft_twiddles :93, mul :133, div :156, syn_div_in_place :190,
// syn_div_expanded_in_place :202, degree_of :242, infer_degree :251) and src/math/quartic.rs
// (eval :6, evaluate_batch :20, interpolate_batch :37, transpose :137).
#pragma once
#include "field.hpp"
#include <array>
#include <algorithm>

namespace orc {

typedef std::vector<u128> vec;
typedef std::array<u128, 4> quad;

// ---- fft.rs -----------------------------------------------------------------------------------
static inline size_t permute_index(size_t size, size_t index) {   // fft.rs:83
    if (size == 1) return 0;
    int bits = __builtin_ctzll((unsigned long long)size);
    size_t r = 0;
    for (int i = 0; i < bits; i++) r |= ((index >> i) & 1) << (bits - 1 - i);
    return r;
}
static inline void permute(u128* v, size_t n) {                   // fft.rs:71
    for (size_t i = 0; i < n; i++) {
        size_t j = permute_index(n, i);
        if (j > i) std::swap(v[i], v[j]);
    }
}
static inline void permute(vec& v) { permute(v.data(), v.size()); }

static const size_t FFT_MAX_LOOP = 256;                           // fft.rs:7

// in-place recursive radix-2 transform, permuted (bit-reversed) output          fft.rs:16-56
static void fft_in_place(u128* values, size_t len, const u128* twiddles, size_t count, size_t stride, size_t offset) {
    size_t size = len / stride;
    if (size > 2) {
        if (stride == count && count < FFT_MAX_LOOP) {
            fft_in_place(values, len, twiddles, 2 * count, 2 * stride, offset);
        } else {
            fft_in_place(values, len, twiddles, count, 2 * stride, offset);
            fft_in_place(values, len, twiddles, count, 2 * stride, offset + stride);
        }
    }
    for (size_t o = offset; o < offset + count; o++) {            // butterfly, fft.rs:92
        u128 t = values[o];
        values[o] = add(t, values[o + stride]);
        values[o + stride] = sub(t, values[o + stride]);
    }
    size_t last_offset = offset + size * stride;
    size_t i = 1;
    for (size_t o = offset + 2 * stride; o < last_offset; o += 2 * stride, i++) {
        for (size_t j = o; j < o + count; j++) {                  // butterfly_twiddle, fft.rs:101
            u128 t = values[j];
            u128 m = mul(values[j + stride], twiddles[i]);
            values[j] = add(t, m);
            values[j + stride] = sub(t, m);
        }
    }
}

static inline vec get_twiddles(u128 root, size_t size) {          // fft.rs:58
    assert(exp(root, (u128)size) == 1);
    vec tw = get_power_series(root, size / 2);
    permute(tw);
    return tw;
}
static inline vec get_inv_twiddles(u128 root, size_t size) {      // fft.rs:66
    return get_twiddles(exp(root, (u128)(size - 1)), size);
}

// ---- polynom.rs -------------------------------------------------------------------------------
static inline u128 poly_eval(const u128* p, size_t n, u128 x) {   // polynom.rs:9
    u128 y = 0, pw = 1;
    for (size_t i = 0; i < n; i++) {
        y = add(y, mul(p[i], pw));
        pw = mul(pw, x);
    }
    return y;
}
static inline u128 poly_eval(const vec& p, u128 x) { return poly_eval(p.data(), p.size(), x); }

static inline void eval_fft_twiddles(u128* p, size_t n, const vec& twiddles, bool unpermute) {   // polynom.rs:34
    assert(n == twiddles.size() * 2);
    fft_in_place(p, n, twiddles.data(), 1, 1, 0);
    if (unpermute) permute(p, n);
}
static inline void interpolate_fft_twiddles(u128* v, size_t n, const vec& inv_twiddles, bool unpermute) {  // polynom.rs:93
    fft_in_place(v, n, inv_twiddles.data(), 1, 1, 0);
    u128 inv_len = inv((u128)n);
    for (size_t i = 0; i < n; i++) v[i] = mul(v[i], inv_len);
    if (unpermute) permute(v, n);
}
static inline void eval_fft(vec& p) {                             // polynom.rs:23
    eval_fft_twiddles(p.data(), p.size(), get_twiddles(get_root_of_unity(p.size()), p.size()), true);
}
static inline void interpolate_fft(vec& v) {                      // polynom.rs:81
    interpolate_fft_twiddles(v.data(), v.size(), get_inv_twiddles(get_root_of_unity(v.size()), v.size()), true);
}

static inline size_t degree_of(const vec& p) {                    // polynom.rs:242
    for (size_t i = p.size(); i-- > 0;) if (p[i] != 0) return i;
    return 0;
}
static inline size_t infer_degree(const vec& evaluations) {       // polynom.rs:251
    vec p = evaluations;
    interpolate_fft(p);
    return degree_of(p);
}

static inline vec poly_mul(const vec& a, const vec& b) {          // polynom.rs:133
    vec r(a.size() + b.size() - 1, 0);
    for (size_t i = 0; i < a.size(); i++)
        for (size_t j = 0; j < b.size(); j++) r[i + j] = add(r[i + j], mul(a[i], b[j]));
    return r;
}
static inline vec poly_div(const vec& a_in, const vec& b) {       // polynom.rs:156 (remainder ignored)
    size_t apos = degree_of(a_in), bpos = degree_of(b);
    assert(apos >= bpos);
    vec a = a_in;
    vec result(apos - bpos + 1, 0);
    for (size_t i = result.size(); i-- > 0;) {
        u128 quot = div(a[apos], b[bpos]);
        result[i] = quot;
        for (size_t j = bpos; j-- > 0;) a[i + j] = sub(a[i + j], mul(b[j], quot));
        apos--;
    }
    return result;
}

static inline void syn_div_in_place(u128* a, size_t n, u128 b) {  // polynom.rs:190
    u128 c = 0;
    for (size_t i = n; i-- > 0;) {
        u128 t = add(a[i], mul(b, c));
        a[i] = c;
        c = t;
    }
}
static inline void syn_div_in_place(vec& a, u128 b) { syn_div_in_place(a.data(), a.size(), b); }

// divide by (x^degree - 1) / prod (x - exceptions[i])                            polynom.rs:202-236
static inline void syn_div_expanded_in_place(vec& a, size_t degree, const vec& exceptions) {
    vec result(a.begin(), a.end());
    result.reserve(a.size() + exceptions.size());
    size_t degree_offset = a.size() - degree;
    for (size_t i = degree_offset; i-- > 0;) result[i] = add(result[i], result[i + degree]);
    for (u128 e : exceptions) {
        u128 ne = neg(e);
        result.push_back(0);   // the reference extends into zero-filled spare capacity (polynom.rs:205,221)
        u128 next_term = result[0];
        result[0] = 0;
        for (size_t i = 0; i + 1 < result.size(); i++) {
            result[i] = add(result[i], mul(next_term, ne));
            std::swap(next_term, result[i + 1]);
        }
    }
    size_t keep = degree_offset + exceptions.size();
    for (size_t i = 0; i < keep; i++) a[i] = result[degree + i];
    for (size_t i = keep; i < a.size(); i++) a[i] = 0;
}

// Lagrange interpolation (used by the FRI remainder check)                       polynom.rs:47-75, 260-276
static inline vec poly_interpolate(const vec& xs, const vec& ys) {
    size_t m = xs.size();
    // get_zero_roots
    vec roots(m + 1);
    size_t n = m;
    roots[n] = 1;
    for (size_t i = 0; i < m; i++) {
        n -= 1;
        roots[n] = 0;
        for (size_t j = n; j < m; j++) roots[j] = sub(roots[j], mul(roots[j + 1], xs[i]));
    }
    std::vector<vec> numerators;
    for (size_t i = 0; i < m; i++) numerators.push_back(poly_div(roots, vec{neg(xs[i]), 1}));
    vec denominators(m), inv_den(m);
    for (size_t i = 0; i < m; i++) denominators[i] = poly_eval(numerators[i], xs[i]);
    inv_many_fill(denominators.data(), inv_den.data(), m);
    vec result(m, 0);
    for (size_t i = 0; i < m; i++) {
        u128 y_slice = mul(ys[i], inv_den[i]);
        for (size_t j = 0; j < m; j++)
            if (numerators[i][j] != 0 && ys[i] != 0) result[j] = add(result[j], mul(numerators[i][j], y_slice));
    }
    return result;
}

// ---- quartic.rs -------------------------------------------------------------------------------
static inline u128 quartic_eval(const quad& p, u128 x) {          // quartic.rs:6
    u128 y = add(p[0], mul(p[1], x));
    u128 x2 = mul(x, x);
    y = add(y, mul(p[2], x2));
    u128 x3 = mul(x2, x);
    y = add(y, mul(p[3], x3));
    return y;
}
static inline vec quartic_evaluate_batch(const std::vector<quad>& polys, u128 x) {   // quartic.rs:20
    vec r(polys.size());
    for (size_t i = 0; i < polys.size(); i++) r[i] = quartic_eval(polys[i], x);
    return r;
}
static inline std::vector<quad> quartic_interpolate_batch(const std::vector<quad>& xs, const std::vector<quad>& ys) {  // quartic.rs:37
    size_t n = xs.size();
    std::vector<quad> equations(n * 4);
    vec inverses(n * 4);
    for (size_t i = 0, j = 0; i < n; i++, j += 4) {
        const quad& x = xs[i];
        u128 x01 = mul(x[0], x[1]), x02 = mul(x[0], x[2]), x03 = mul(x[0], x[3]);
        u128 x12 = mul(x[1], x[2]), x13 = mul(x[1], x[3]), x23 = mul(x[2], x[3]);
        equations[j] = {mul(neg(x12), x[3]), add(add(x12, x13), x23), sub(sub(neg(x[1]), x[2]), x[3]), 1};
        inverses[j] = quartic_eval(equations[j], x[0]);
        equations[j + 1] = {mul(neg(x02), x[3]), add(add(x02, x03), x23), sub(sub(neg(x[0]), x[2]), x[3]), 1};
        inverses[j + 1] = quartic_eval(equations[j + 1], x[1]);
        equations[j + 2] = {mul(neg(x01), x[3]), add(add(x01, x03), x13), sub(sub(neg(x[0]), x[1]), x[3]), 1};
        inverses[j + 2] = quartic_eval(equations[j + 2], x[2]);
        equations[j + 3] = {mul(neg(x01), x[2]), add(add(x01, x02), x12), sub(sub(neg(x[0]), x[1]), x[2]), 1};
        inverses[j + 3] = quartic_eval(equations[j + 3], x[3]);
    }
    vec invd(n * 4);
    inv_many_fill(inverses.data(), invd.data(), n * 4);
    std::vector<quad> result(n);
    for (size_t i = 0, j = 0; i < n; i++, j += 4) {
        const quad& y = ys[i];
        quad r = {0, 0, 0, 0};
        for (int k = 0; k < 4; k++) {
            u128 inv_y = mul(y[k], invd[j + k]);
            for (int c = 0; c < 4; c++) r[c] = add(r[c], mul(inv_y, equations[j + k][c]));
        }
        result[i] = r;
    }
    return result;
}
static inline std::vector<quad> quartic_transpose(const u128* v, size_t len, size_t stride) {   // quartic.rs:137
    assert(len % (4 * stride) == 0);
    size_t rows = len / (4 * stride);
    std::vector<quad> r(rows);
    for (size_t i = 0; i < rows; i++)
        r[i] = {v[i * stride], v[(i + rows) * stride], v[(i + 2 * rows) * stride], v[(i + 3 * rows) * stride]};
    return r;
}
static inline std::vector<quad> quartic_transpose(const vec& v, size_t stride) { return quartic_transpose(v.data(), v.size(), stride); }

}  // namespace orc
