"""Pins the oracle's field / FFT / polynomial / quartic layer against the reference's own test vectors
(/root/reference/src/math/{field,fft,polynom,quartic}.rs test modules) and against Python big integers."""
import random

import numpy as np

P = 2**128 - 45 * 2**40 + 1


def test_field_literals(oracle):
    O = oracle
    # field.rs:346-402
    assert O.add(2, 3) == 5 and O.add(P - 1, 1) == 0 and O.add(P - 1, 2) == 1
    assert O.sub(5, 3) == 2 and O.sub(3, 5) == P - 2
    assert O.mul(5, 3) == 15 and O.mul(P - 1, P - 1) == 1 and O.mul(P - 1, 2) == P - 2 and O.mul(P - 1, 4) == P - 4
    assert O.mul((P + 1) // 2, 2) == 1
    assert O.inv(1) == 1 and O.inv(0) == 0                     # field.rs:424-436
    assert O.exp(0, 5) == 0 and O.exp(7, 0) == 1               # field.rs:201-203


def test_field_random_vs_bigint(oracle):
    O = oracle
    rnd = random.Random(7)
    a = [rnd.randrange(P) for _ in range(3000)] + [0, 1, P - 1, P - 2, 2**64, 2**64 - 1, 2**127]
    b = [rnd.randrange(P) for _ in range(3000)] + [P - 1, P - 1, P - 1, 2, 2**64, 2**64 + 1, 2**127 + 1]
    A, B = O.to_arr(a), O.to_arr(b)
    assert O.to_ints(O.field_op("add", A, B)) == [(x + y) % P for x, y in zip(a, b)]
    assert O.to_ints(O.field_op("sub", A, B)) == [(x - y) % P for x, y in zip(a, b)]
    assert O.to_ints(O.field_op("mul", A, B)) == [(x * y) % P for x, y in zip(a, b)]
    inv = O.to_ints(O.field_op("inv", A[:200]))
    assert all((x * y) % P == (1 if x else 0) for x, y in zip(a[:200], inv))
    assert O.to_ints(O.field_op("exp", A[:50], B[:50])) == [pow(x, y, P) for x, y in zip(a[:50], b[:50])]
    vals = a[:100] + [0, 0] + a[100:120]
    got = O.to_ints(O.inv_many(O.to_arr(vals)))               # field.rs:173 (zeros stay zero)
    assert got == [pow(v, P - 2, P) if v else 0 for v in vals]


def test_root_of_unity(oracle):
    O = oracle
    root_40 = O.root_of_unity(2**40)                           # field.rs:438-448
    assert root_40 == 23953097886125630542083529559205016746
    assert O.exp(root_40, 2**40) == 1
    root_39 = O.root_of_unity(2**39)
    assert root_39 == O.exp(root_40, 2) and O.exp(root_39, 2**39) == 1
    assert O.to_ints(O.power_series(3, 5)) == [1, 3, 9, 27, 81]


def naive_eval(poly, x):
    y, pw = 0, 1
    for c in poly:
        y = (y + c * pw) % P
        pw = pw * x % P
    return y


def test_fft_matches_naive_evaluation(oracle):
    O = oracle
    rnd = random.Random(3)
    for n in (4, 8, 16, 1024):                                 # fft.rs:116-157
        p = list(range(1, n + 1)) if n <= 16 else [rnd.randrange(P) for _ in range(n)]
        g = O.root_of_unity(n)
        xs = [pow(g, i, P) for i in range(n)]
        expected = [naive_eval(p, x) for x in xs]
        assert O.to_ints(O.fft_eval(O.to_arr(p))) == expected
        assert O.to_ints(O.fft_interpolate(O.to_arr(expected))) == p


def test_poly_eval_literals(oracle):
    O = oracle
    x = 11269864713250585702                                   # polynom.rs:286-313
    poly = [384863712573444386, 7682273369345308472, 13294661765012277990, 16234810094004944758]
    assert O.poly_eval(O.to_arr(poly[:1]), x) == poly[0]
    for k in (2, 3, 4):
        assert O.poly_eval(O.to_arr(poly[:k]), x) == naive_eval(poly[:k], x)


def test_syn_div(oracle):
    O = oracle
    poly = O.to_ints(O.poly_mul(O.to_arr([2, 1]), O.to_arr([3, 1])))          # polynom.rs:453-461
    got = O.to_ints(O.syn_div(O.to_arr(poly), P - 3))
    assert got[:2] == O.to_ints(O.poly_div(O.to_arr(poly), O.to_arr([3, 1]))) and got[2] == 0


def test_syn_div_expanded(oracle):
    O = oracle
    ys = [0, 1, 2, 3, 0, 5, 6, 7, 0, 9, 10, 11, 12, 13, 14, 15]               # polynom.rs:466-490
    poly = O.to_ints(O.fft_interpolate(O.to_arr(ys)))
    root = O.root_of_unity(16)
    d12 = pow(root, 12, P)
    z_poly = O.to_ints(O.poly_div(O.to_arr([P - 1, 0, 0, 0, 1]), O.to_arr([(P - d12) % P, 1])))
    result = O.to_ints(O.syn_div_expanded(O.to_arr(poly), 4, [d12]))
    expected = O.to_ints(O.poly_div(O.to_arr(poly), O.to_arr(z_poly)))
    while result and result[-1] == 0:
        result.pop()
    assert result == expected
    back = O.to_ints(O.poly_mul(O.to_arr(expected), O.to_arr(z_poly)))
    while back and back[-1] == 0:
        back.pop()
    assert back == poly


def test_infer_degree(oracle):
    O = oracle
    for size in (16, 32):                                      # polynom.rs:502-515
        ev = O.fft_eval(O.to_arr([1, 2, 3, 4] + [0] * (size - 4)))
        assert O.infer_degree(ev) == 3


def test_quartic(oracle):
    O = oracle
    v = O.to_arr(list(range(1, 17)))
    assert O.to_ints(O.quartic_transpose(v, 1)) == [[1, 5, 9, 13], [2, 6, 10, 14], [3, 7, 11, 15], [4, 8, 12, 16]]   # quartic.rs:219-227
    assert O.to_ints(O.quartic_transpose(v, 2)) == [[1, 5, 9, 13], [3, 7, 11, 15]]
    r = O.root_of_unity(16)                                    # quartic.rs:177-191: interpolate_batch == Lagrange
    xs = np.ascontiguousarray(O.power_series(r, 16)).reshape(4, 4, 2)
    ys = O.to_arr([[1, 2, 3, 4], [5, 6, 7, 8], [9, 10, 11, 12], [13, 14, 15, 16]])
    got = O.to_ints(O.quartic_interpolate_batch(xs, ys))
    for i in range(4):
        assert got[i] == O.to_ints(O.poly_interpolate(xs[i], ys[i]))
        for k in range(4):
            assert naive_eval(got[i], O.to_ints(xs[i])[k]) == O.to_ints(ys[i])[k]
    polys = [[7956382178997078105, 6172178935026293282, 5971474637801684060, 16793452009046991148],    # quartic.rs:193-210
             [7956382178997078109, 15205743380705406848, 12475269242634339237, 194846859619262948]]
    x = 987654321987654321987654321
    assert O.to_ints(O.quartic_evaluate_batch(O.to_arr(polys), x)) == [naive_eval(p, x) for p in polys]
