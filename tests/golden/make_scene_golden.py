"""Writes tests/golden/scene_golden.json: for each scene of tests/test_scene_load.py the digest of what the product's
loaders produce, recorded only after it was found identical -- listing, triangles, texels, sky -- to the output of the
REFERENCE'S OWN loaders (oracle/_ref/libref_scene.so = /root/reference/Src compiled verbatim). Runs in the build
container only (it needs the reference mount); the digests are what travels.

    python tests/golden/make_scene_golden.py
"""
import json, os, pathlib, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gpu_raytracer_amd as grt
from oracle import binding as oracle
import test_scene_load as t


def main():
    assert oracle.ref_scene_lib() is not None, "oracle/_ref/libref_scene.so missing: run `make -C oracle ref` where /root/reference exists"
    golden = {}
    with tempfile.TemporaryDirectory() as d:
        d = pathlib.Path(d)
        sky = t.write_sky(d / "sky.hdr")
        golden["cornellbox"] = t.assert_same_scene(grt, oracle, t.posix_copy_of(grt, "cornellbox", d), sky)[0]
        sponza = t.posix_copy_of(grt, "sponza", d)
        for bc in (1, 0):
            golden["sponza_bc%d" % bc] = t.assert_same_scene(grt, oracle, sponza, sky, enable_block_compression=bc)[0]
        (d / "f").mkdir()
        t.write_feature_scene(d / "f"); t.write_sky(d / "f" / "sky.hdr", seed=2)
        os.chdir(d / "f")                 # relative names, as in the test: a hair file's ribbon angle is seeded from its file name
        golden["features"] = t.assert_same_scene(grt, oracle, "features.xml", "sky.hdr")[0]
        os.chdir(ROOT)
    json.dump(golden, open(t.GOLDEN_PATH, "w"), indent=1, sort_keys=True)
    print("wrote", t.GOLDEN_PATH, golden)


if __name__ == "__main__":
    main()
