"""Generates tests/golden/blocks/*: small inputs and the bytes third-party libraries produce
for them (liblz4 1.9.3 LZ4_compress_default via ctypes, libzstd and snappy via pyarrow).
Run in the build container; the fixtures (data only) are committed.  The reference itself
(Rust) cannot be run here, so these pin the block formats the reference delegates to
liblz4 / libzstd / snap (src/compression/basic.rs:87-152)."""
import ctypes as C
import json
import os

import numpy as np
import pyarrow as pa

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "blocks")


def inputs():
    rng = np.random.default_rng(7)
    words = [b"w%d" % i + b"x" * (i % 9) for i in range(200)]
    return {
        "empty": np.zeros(0, np.uint8),
        "tiny": np.frombuffer(b"abc", np.uint8),
        "thirteen_a": np.frombuffer(b"a" * 13, np.uint8),
        "abcd_x500": np.frombuffer(b"abcd" * 500, np.uint8),
        "random_3k": rng.integers(0, 256, 3000, dtype=np.uint8),
        "low_entropy_8k": rng.integers(0, 4, 8000, dtype=np.uint8),
        "i32_runs": np.repeat(rng.integers(0, 50, 300), 7).astype(np.int32).view(np.uint8),
        "f64_small_set": (rng.integers(0, 16, 1500) * 0.25).astype(np.float64).view(np.uint8),
        "offsets_i32": np.cumsum(rng.integers(0, 6, 1200)).astype(np.int32).view(np.uint8),
        "zipf_words": np.frombuffer(b"".join(words[i] for i in rng.zipf(1.3, 1500) % 200), np.uint8),
        "zeros_70k": np.zeros(70000, np.uint8),  # > 64 KiB: LZ4 switches to the 32-bit hash table
    }


def main():
    os.makedirs(OUT, exist_ok=True)
    lz = C.CDLL("liblz4.so.1")
    lz.LZ4_compress_default.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    lz.LZ4_compressBound.argtypes = [C.c_int]
    lz.LZ4_versionString.restype = C.c_char_p
    index = {"liblz4": lz.LZ4_versionString().decode(), "pyarrow": pa.__version__, "cases": []}
    for name, a in inputs().items():
        a = np.ascontiguousarray(a)
        a.tofile(os.path.join(OUT, name + ".raw"))
        cap = lz.LZ4_compressBound(a.size)
        dst = np.zeros(cap, np.uint8)
        n = lz.LZ4_compress_default(a.ctypes.data, dst.ctypes.data, a.size, cap)
        dst[:n].tofile(os.path.join(OUT, name + ".lz4"))
        for lvl in (1, 3):
            z = pa.Codec("zstd", compression_level=lvl).compress(a.tobytes(), asbytes=True)
            open(os.path.join(OUT, "%s.zstd%d" % (name, lvl)), "wb").write(z)
        s = pa.Codec("snappy").compress(a.tobytes(), asbytes=True)
        open(os.path.join(OUT, name + ".snappy"), "wb").write(s)
        index["cases"].append({"name": name, "size": int(a.size)})
    json.dump(index, open(os.path.join(OUT, "index.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
