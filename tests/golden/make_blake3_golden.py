#!/usr/bin/env python3
"""Generates tests/golden/blake3_golden.json with the OFFICIAL BLAKE3 C implementation that ships inside
/opt/rocm/lib/llvm/lib/libclang-cpp.so (llvm_blake3_hasher_init/update/finalize).  Run in the build container:

    python tests/golden/make_blake3_golden.py

The reference hashes with the third-party `blake3` crate (src/crypto/hash.rs:205-209) and holds no BLAKE3 test vector
of its own, so these digests pin the oracle's (and through it the HIP kernels') BLAKE3 for the exact input shapes on the
prover path: trace rows of W*16 bytes (W = 1..127 registers: 16..2032 B, i.e. up to two chunks), 64-byte Merkle node
pairs / FRI rows / proof-of-work inputs, and the concatenation of FRI layer roots.
"""
import ctypes, json, os

LIB = "/opt/rocm/lib/llvm/lib/libclang-cpp.so"

def pattern(n, salt):
    return bytes(((i * 7 + salt * 13 + (i >> 8) * 31 + 3) % 251) for i in range(n))

def main():
    lib = ctypes.CDLL(LIB)
    def h(data):
        st = ctypes.create_string_buffer(4096)
        lib.llvm_blake3_hasher_init(st)
        lib.llvm_blake3_hasher_update(st, data, ctypes.c_size_t(len(data)))
        out = ctypes.create_string_buffer(32)
        lib.llvm_blake3_hasher_finalize(st, out, ctypes.c_size_t(32))
        return out.raw.hex()
    assert h(b"") == "af1349b9f5f9a1a6a0404dea36dcc9499bcb25c9adc112b7cc9a93cae41f3262"   # published empty-input vector
    lengths = [0, 1, 16, 63, 64, 65, 127, 128, 129, 256, 272, 320, 384, 512, 1023, 1024, 1025, 1040, 1088, 2032, 2048, 2049, 3072, 4097]
    cases = []
    for salt, n in enumerate(lengths):
        cases.append({"len": n, "salt": salt, "digest": h(pattern(n, salt))})
    # all row widths W = 1..127 (W*16 bytes)
    for w in range(1, 128):
        cases.append({"len": 16 * w, "salt": 100 + w, "digest": h(pattern(16 * w, 100 + w))})
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "blake3_golden.json")
    json.dump({"generator": "tests/golden/make_blake3_golden.py", "source": "llvm_blake3 (official C implementation) in libclang-cpp.so, ROCm 7.2",
               "pattern": "byte i = (i*7 + salt*13 + (i>>8)*31 + 3) % 251", "cases": cases}, open(out, "w"), indent=0)
    print("wrote", out, len(cases), "cases")

if __name__ == "__main__":
    main()
