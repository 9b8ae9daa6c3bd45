#!/usr/bin/env python3
"""Extracts the reference's own golden vectors for the path into small JSON fixtures.

Run in the build container (needs /root/reference, which does not exist on the GPU box):
    python tests/golden/make_golden.py

Outputs (committed):
  sfmt_seed4321.json   -- first 64-bit outputs of Random(4321), table in
                          src/tests/test_random.cpp:436-466 (checked there at :468-472)
  clipped_aabb.json    -- the five Triangle::getClippedAABB known answers of
                          src/tests/test_kd.cpp:34-84
  test_bsdf_subset.json-- the BSDF configurations of data/tests/test_bsdf.xml that are on the
                          path (diffuse, twosided(diffuse), dielectric, roughconductor ...)
"""
import json, os, re, sys
import xml.etree.ElementTree as ET

REF = os.environ.get("PHIP_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))


def sfmt():
    src = open(os.path.join(REF, "src/tests/test_random.cpp")).read()
    start = src.index("static const uint64_t reference[]")
    end = src.index("};", start)
    words = re.findall(r"0x([0-9a-fA-F]{16})ULL", src[start:end])
    seed = int(re.search(r"new Random\((\d+)\)", src[end:end + 400]).group(1))
    json.dump({"source": "src/tests/test_random.cpp:436-472", "seed": seed, "words_hex": words},
              open(os.path.join(HERE, "sfmt_seed4321.json"), "w"), indent=0)
    print("sfmt: seed", seed, len(words), "words")


def clipped():
    # transcribed from the assertions of test01_sutherlandHodgman (test_kd.cpp:34-84)
    tri = [0, 0, 0, 1, 0, 0, 1, 1, 0]
    cases = [
        {"box": [0, .5, -1, 1, 1, 1], "valid": True, "min": [.5, .5, 0], "max": [1, 1, 0]},
        {"box": [2, 2, 2, 3, 3, 3], "valid": False},
        {"box": [-1, -1, -1, 1, 1, 1], "valid": True, "min": [0, 0, 0], "max": [1, 1, 0]},
        {"box": [-100, -100, 0, 100, 100, 0], "valid": True, "min": [0, 0, 0], "max": [1, 1, 0]},
        {"box": [0, 1, 0, 1, 2, 0], "valid": True, "min": [1, 1, 0], "max": [1, 1, 0]},
    ]
    src = open(os.path.join(REF, "src/tests/test_kd.cpp")).read()
    # sanity: the numbers above must literally appear in the reference test
    for needle in ["Point(0, .5, -1)", "Point(2, 2, 2)", "Point(-100,-100, 0)", "Point(0,1, 0)", "Point(.5, .5, 0)"]:
        assert needle in src, needle
    json.dump({"source": "src/tests/test_kd.cpp:34-84", "triangle": tri, "cases": cases},
              open(os.path.join(HERE, "clipped_aabb.json"), "w"), indent=0)
    print("clipped_aabb:", len(cases), "cases")


def bsdfs():
    root = ET.parse(os.path.join(REF, "data/tests/test_bsdf.xml")).getroot()
    out = []

    def conv(b):
        t = b.get("type")
        d = {"type": t, "params": {}}
        for c in b:
            if c.tag in ("float", "string", "boolean", "spectrum", "rgb", "integer"):
                d["params"][c.get("name")] = c.get("value")
            elif c.tag == "bsdf":
                d.setdefault("nested", []).append(conv(c))
        return d

    for b in root.findall("bsdf"):
        t = b.get("type")
        if t in ("diffuse", "dielectric", "roughconductor", "twosided"):
            c = conv(b)
            if t == "twosided" and any(n["type"] not in ("diffuse", "roughconductor") for n in c.get("nested", [])):
                continue
            out.append(c)
    json.dump({"source": "data/tests/test_bsdf.xml", "bsdfs": out},
              open(os.path.join(HERE, "test_bsdf_subset.json"), "w"), indent=1)
    print("test_bsdf subset:", [o["type"] for o in out])


if __name__ == "__main__":
    sfmt(); clipped(); bsdfs()
