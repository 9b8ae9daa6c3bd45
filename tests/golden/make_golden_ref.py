#!/usr/bin/env python3
"""Renders the pin scenes of tests/ref_scenes.py with the REFERENCE ITSELF (oracle/_ref: /root/reference compiled in
place by oracle/Makefile.ref) and commits what came out as a fixture:

    tests/golden/ref_renders.npz
        <case>/samples     per-sample (R,G,B,alpha) of the reference's MIPathTracer / MIDirectIntegrator ::Li,
                           `independent` sampler (SFMT19937, one clone), one image block, row-major pixels
        <case>/film        the reference's ImageBlock accumulator (R,G,B,alpha,weight) after every ImageBlock::put
        mip/<scene>/<key>/<level>   the MIP pyramids the reference's own TMIPMap built for the envmap / texture scenes

Run in the build container (needs /root/reference):   python tests/golden/make_golden_ref.py
tests/test_golden.py::test_reference_render_fixture replays the cases on the oracle anywhere."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

import ref_scenes as RS                                   # noqa: E402
from oracle import oracle_ffi as O, ref_ffi as R          # noqa: E402


def main():
    R.build(); O.build(libm=True)
    gauss = O.gaussian_filter(0.5, libm=True)
    out = {}
    for name, build, kw in RS.CASES:
        def mip(key, image, kind, wrap_u="repeat", wrap_v=None, filter_type="ewa", max_anisotropy=None):
            levels = R.RefMip(image, kind=kind, wrap_u=wrap_u, wrap_v=wrap_v, filter_type=filter_type, max_anisotropy=max_anisotropy).levels
            for l, a in enumerate(levels):
                out["mip/%s/%s/%d" % (build.__name__, key, l)] = a
            return levels
        desc = build(gauss, mip).desc()
        rs = R.RefScene(desc)
        film, samples = rs.render(RS.params(kw))
        out[name + "/samples"] = samples
        out[name + "/film"] = film
        print("%-22s %s mean %.5f" % (name, samples.shape, samples[..., :3].mean()))
    path = os.path.join(HERE, "ref_renders.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
