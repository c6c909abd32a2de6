#!/usr/bin/env python3
"""Which compute units a HIP CU mask enables on this part (hipExtStreamCreateWithCUMask): for a few masks, launch 4096
spinning workgroups on a masked stream and collect (XCD, shader engine, CU) of each.  Answers: are mask bits dealt
round-robin over the XCDs (bit i -> XCD i % 8), or laid out XCD by XCD?  scone_set_cu_reserve and the copy stream of the
pinned-host prefetch choose their masks by this.  One JSON line."""
import ctypes
import json
import os
from collections import Counter

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    lib = ctypes.CDLL(os.path.join(HERE, "standin", "libtransport_standin.so"))
    lib.standin_mask_probe.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    n_blocks, words = 4096, 8

    def probe(bits):
        out = (ctypes.c_uint32 * (2 * n_blocks))()
        if bits is None:
            rc = lib.standin_mask_probe(None, 0, n_blocks, 2000, out)
        else:
            m = (ctypes.c_uint32 * words)()
            for b in bits:
                m[b >> 5] |= 1 << (b & 31)
            rc = lib.standin_mask_probe(m, words, n_blocks, 2000, out)
        if rc:
            return {"error": rc}
        cus = Counter()
        for i in range(n_blocks):
            xcc, hw = out[2 * i] & 0xF, out[2 * i + 1]
            cus[(xcc, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15)] += 1
        per_xcd = Counter(k[0] for k in cus)
        return {"distinct_cus": len(cus), "cus_per_xcd": [per_xcd.get(x, 0) for x in range(8)],
                "workgroups_per_xcd": [sum(v for k, v in cus.items() if k[0] == x) for x in range(8)]}

    res = {"no_mask": probe(None), "all_256_bits": probe(range(256)), "top_8_bits_clear": probe(range(248)),
           "top_64_bits_clear": probe(range(192)), "bits_0_to_31": probe(range(32)), "bits_multiple_of_8": probe(range(0, 256, 8)),
           "bits_0_8_16_24": probe([0, 8, 16, 24]), "bits_0_to_7": probe(range(8))}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
