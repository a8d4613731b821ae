#!/usr/bin/env python3
"""Generates tests/golden/chacha20_golden.json with OpenSSL's chacha20 (keystream of an all-zero plaintext, 16-byte
all-zero IV = 32-bit block counter 0 + 96-bit nonce 0, which coincides with the original 64-bit-counter / 64-bit-stream
layout used by rand_chacha's ChaCha20Rng for the first 2^32 blocks).  Run in the build container:

    python tests/golden/make_chacha_golden.py

The reference derives every Fiat-Shamir challenge from rand 0.7's StdRng (= ChaCha20) seeded with a 32-byte Merkle
root (src/math/field.rs:264-275); the crate is not under /root/reference and the reference has no test with concrete
draws, so this pins the ChaCha20 core only; the Uniform<u128> sampling rule stays a restatement (oracle/prng.hpp).
"""
import json, os, subprocess

def keystream(key_hex, nbytes):
    p = subprocess.run(["openssl", "enc", "-chacha20", "-K", key_hex, "-iv", "00" * 16], input=bytes(nbytes), capture_output=True, check=True)
    return p.stdout.hex()

def main():
    keys = ["00" * 32, "000102030405060708090a0b0c0d0e0f101112131415161718191a1b1c1d1e1f", "ff" * 32,
            "".join("%02x" % ((i * 37 + 11) % 256) for i in range(32))]
    cases = [{"seed": k, "keystream": keystream(k, 512)} for k in keys]
    assert cases[0]["keystream"].startswith("76b8e0ada0f13d90405d6ae55386bd28bdd219b8a08ded1aa836efcc8b770dc7")   # RFC 7539 A.1 #1
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "chacha20_golden.json")
    json.dump({"generator": "tests/golden/make_chacha_golden.py", "source": "OpenSSL 3.0.2 `openssl enc -chacha20`", "cases": cases}, open(out, "w"), indent=0)
    print("wrote", out)

if __name__ == "__main__":
    main()
