"""Pins the oracle's BLAKE3, ChaCha20 and Merkle tree against committed golden vectors (tests/golden/*.json, generated
from the official BLAKE3 C code and from OpenSSL) and against the reference's Merkle structure tests
(/root/reference/src/crypto/merkle.rs:321-518, restated with BLAKE3 as the node hash)."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def pattern(n, salt):
    return bytes(((i * 7 + salt * 13 + (i >> 8) * 31 + 3) % 251) for i in range(n))


def test_blake3_golden(oracle):
    cases = json.load(open(os.path.join(GOLDEN, "blake3_golden.json")))["cases"]
    assert len(cases) > 100
    for c in cases:
        assert oracle.blake3(pattern(c["len"], c["salt"])).hex() == c["digest"], c["len"]


def test_chacha20_golden(oracle):
    for c in json.load(open(os.path.join(GOLDEN, "chacha20_golden.json")))["cases"]:
        ks = bytes.fromhex(c["keystream"])
        words = oracle.chacha20_words(bytes.fromhex(c["seed"]), len(ks) // 4)
        assert words.astype("<u4").tobytes() == ks


def test_uniform_sampling_properties(oracle):
    # rand 0.7 Uniform<u128>(0..M): every draw is a canonical field element; draws are deterministic in the seed
    O = oracle
    seed = bytes(range(32))
    a = O.to_ints(O.prng_vector(seed, 600))
    assert all(0 <= v < O.P for v in a) and len(set(a)) == 600
    assert O.to_ints(O.prng_vector(seed, 10)) == a[:10]
    # first draw = floor(v * M / 2^128) of the first accepted 128-bit word pair (widening-multiply rule)
    w = [int(x) for x in O.chacha20_words(seed, 8)]
    v = w[0] | (w[1] << 32) | (w[2] << 64) | (w[3] << 96)
    prod = v * O.P
    if (prod & (2**128 - 1)) <= O.P - 1:
        assert a[0] == prod >> 128
    pos = O.query_positions(seed, 2**15, 32, 50)
    assert len(pos) == 50 and len(set(pos)) == 50 and all(p % 32 != 0 and p < 2**15 for p in pos)


LEAVES = [bytes([(37 * i + 11 * j + 5) % 256 for j in range(32)]) for i in range(8)]


def h2(a, b, O):
    return O.blake3(a + b)


def test_merkle_tree_layout(oracle):
    O = oracle
    nodes = O.merkle_nodes(b"".join(LEAVES[:4]))               # merkle.rs:340-362 new_tree
    assert nodes[:32] == bytes(32)
    root4 = h2(h2(LEAVES[0], LEAVES[1], O), h2(LEAVES[2], LEAVES[3], O), O)
    assert nodes[32:64] == root4
    nodes = O.merkle_nodes(b"".join(LEAVES))
    n = [nodes[32 * i:32 * i + 32] for i in range(8)]
    assert n[4] == h2(LEAVES[0], LEAVES[1], O) and n[7] == h2(LEAVES[6], LEAVES[7], O)
    assert n[2] == h2(n[4], n[5], O) and n[3] == h2(n[6], n[7], O) and n[1] == h2(n[2], n[3], O)


def test_merkle_prove_batch(oracle):
    O = oracle
    L = LEAVES
    leaves = b"".join(L)
    h01, h23, h45, h67 = h2(L[0], L[1], O), h2(L[2], L[3], O), h2(L[4], L[5], O), h2(L[6], L[7], O)
    pr = O.merkle_prove_batch(leaves, [1])                     # merkle.rs:423-491
    assert pr["values"] == [L[1]] and pr["nodes"] == [[L[0], h23, h2(h45, h67, O)]] and pr["depth"] == 3
    pr = O.merkle_prove_batch(leaves, [1, 2])
    assert pr["values"] == [L[1], L[2]] and pr["nodes"] == [[L[0], h2(h45, h67, O)], [L[3]]]
    pr = O.merkle_prove_batch(leaves, [1, 6])
    assert pr["values"] == [L[1], L[6]] and pr["nodes"] == [[L[0], h23], [L[7], h45]]
    pr = O.merkle_prove_batch(leaves, list(range(8)))
    assert pr["values"] == L and pr["nodes"] == [[], [], [], []] and pr["depth"] == 3


def test_merkle_verify_batch(oracle):
    O = oracle
    leaves = b"".join(LEAVES)
    root = O.merkle_nodes(leaves)[32:64]
    def pv(prove_idx, verify_idx):
        return O.merkle_verify_batch(root, verify_idx, O.merkle_prove_batch(leaves, prove_idx, raw=True))
    assert pv([1], [1]) and not pv([1], [2])                   # merkle.rs:493-518
    assert pv([1, 2], [1, 2]) and not pv([1, 2], [1]) and not pv([1, 2], [1, 3]) and not pv([1, 2], [1, 2, 3])
    assert pv([1, 6], [1, 6]) and pv([1, 3, 6], [1, 3, 6]) and pv(list(range(8)), list(range(8)))
    assert pv([6, 1, 3], [6, 1, 3])                            # request order is preserved in `values`


def test_pow(oracle):
    O = oracle
    seed = bytes(range(32))
    digest, nonce = O.pow_find(seed, 8)                        # proof_of_work.rs:4-32
    assert nonce >= 1
    assert O.blake3(seed + nonce.to_bytes(8, "little") + bytes(24)) == digest
    assert int.from_bytes(digest[:8], "little") % 256 == 0
    for k in range(1, nonce):                                  # it is the FIRST satisfying nonce
        d = O.blake3(seed + k.to_bytes(8, "little") + bytes(24))
        assert int.from_bytes(d[:8], "little") % 256 != 0
