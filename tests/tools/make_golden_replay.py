#!/usr/bin/env python3
"""Generate tests/golden/replay_cases.json with the REAL replay reader of the reference
(include/replay.hpp compiled where it lies into oracle/_ref/libref_replay.so, oracle/Makefile).
Runs ONLY where /root/reference is mounted.  The fixture is data: replay-file texts written here
and the ticks the reference's ReplayElements::loadFromFile parses from them (float32 bit patterns)."""
import ctypes as C
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
GOLD = os.path.join(ROOT, "tests", "golden", "replay_cases.json")

CASES = {
    "plain": "0.0 1 2 3 0.1 -0.2\n0.5 4 5 6 0.3 0.4\n",
    "whitespace_and_exponents": "  0.25\t256.5 1.0e2\n200 1e-3 -1.5E+0\n\n\n1 2 3 4 5 6   ",
    "tick_cut_short_at_end": "0 1 2 3 4 5\n1 1 2 3 4\n",
    "malformed_token_stops_reading": "0.0 1 2 3 0.1 -0.2\n0.5 4 5 6\n0.3 0.4\n1.0 7 8 9 x 0\n2.0 1 1 1 1 1\n",
    "number_glued_to_text": "0 1 2 3 4 5\n6 7 8 9 10 11abc 12 13 14 15 16 17\n",
    "empty": "",
    "only_text": "timestamp x y z view_x view_y\n0 1 2 3 4 5\n",
    "precision_and_range": "0.1 0.2 0.30000001 16777217 1e38 1e-45\n3.4e38 -3.4e38 1e39 0 0 0\n1 1 1 1 1 1\n",
    "hex_and_signs": "+1 -2 +3.5 -4.25 0x10 5\n1 1 1 1 1 1\n",
    "nan_inf_words": "0 1 2 3 4 5\nnan 1 2 3 4 5\n1 1 1 1 1 1\n",
    "orbit_like": "".join(f"{i / 30:.4f} {256 + 100 * ((i * 7) % 11 - 5) / 5:.3f} 186.000 {256 - i:.3f} {-0.1 * i:.5f} -0.45\n" for i in range(12)),
}


def main():
    if not os.path.isdir("/root/reference"):
        sys.exit("reference not mounted; the fixture can only be regenerated in the dev container")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"], stdout=subprocess.DEVNULL)
    lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_replay.so"))
    lib.ref_replay_load.restype = C.c_int32
    lib.ref_replay_load.argtypes = [C.c_char_p, C.c_void_p, C.c_int32]
    out = {}
    with tempfile.TemporaryDirectory() as d:
        for name, text in CASES.items():
            path = os.path.join(d, name + ".replay")
            with open(path, "w") as f:
                f.write(text)
            buf = np.zeros((64, 6), np.float32)
            n = lib.ref_replay_load(path.encode(), buf.ctypes.data, 64)
            assert n <= 64
            out[name] = {"text": text, "ticks_f32_bits": buf[:n].view(np.uint32).tolist()}
        buf = np.zeros((4, 6), np.float32)
        assert lib.ref_replay_load(os.path.join(d, "does_not_exist").encode(), buf.ctypes.data, 4) == 0
    with open(GOLD, "w") as f:
        json.dump(out, f, indent=1)
    for k, v in out.items():
        print(f"{k}: {len(v['ticks_f32_bits'])} ticks")


if __name__ == "__main__":
    main()
