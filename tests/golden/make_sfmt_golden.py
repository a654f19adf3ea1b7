"""Extracts the SFMT-19937 golden vector the reference's own test holds -- the `reference[]` table of
/root/reference/src/tests/test_random.cpp:436-501 (TestRandom::test00_validate: `Random(4321)`, every entry == nextULong()) --
into tests/golden/sfmt_reference.json.  DATA only (192 64-bit numbers and the seed); runs in the build container, where
/root/reference exists; the JSON travels to the GPU box, the reference does not.

    python tests/golden/make_sfmt_golden.py
"""
import json
import os
import re

SRC = "/root/reference/src/tests/test_random.cpp"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "sfmt_reference.json")


def main():
    text = open(SRC).read()
    a = text.index("static const uint64_t reference[] = {")
    b = text.index("};", a)
    vals = re.findall(r"0x([0-9a-fA-F]{16})ULL", text[a:b])
    m = re.search(r"new Random\((\d+)\);\s*for \(size_t i = 0; i < array_size\(reference\)", text[b:b + 400])
    seed = int(m.group(1))
    json.dump({"source": "src/tests/test_random.cpp:436-501 (TestRandom::test00_validate)", "generator": "SFMT-19937, Random(seed), nextULong()",
               "seed": seed, "count": len(vals), "values_hex": vals}, open(OUT, "w"), indent=0)
    print("wrote", OUT, len(vals), "values, seed", seed)


if __name__ == "__main__":
    main()
