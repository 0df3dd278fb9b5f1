"""Flattened scenes of the ten OBJ / MTL models the reference ships under resources/cornell-box, as this
repository's host loader (pathtracer_amd/host: JSON scene reader, OBJ / MTL loader, mesh baking, BVH builder)
produces them from resources/scenes/cornell.json with the model's file name put in.  One SceneData .npz per model
under tests/golden/f1/.  The reference's own import path (Assimp) cannot be built here, so these files pin the
loader against itself; the only figures from the real binary are those of CornellBox-Original (36 triangles, 2
area lights, 71 BVH nodes, depth 7), which tests/test_host_cornell_models.py checks as well.

usage (in the build container, /root/reference present):  python tests/golden/make_f1_golden.py
"""
import glob
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from pathtracer_amd import host  # noqa: E402

RES = "/root/reference/resources/"
OUT = os.path.join(ROOT, "tests", "golden", "f1")


def load_model(obj_name):
    """cornell.json with another model file (and without its log file)."""
    text = open(RES + "scenes/cornell.json").read()
    text = text.replace("CornellBox-Original.obj", obj_name).replace('"LogFile": "cornell.txt",', "")
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "scene.json")
        open(path, "w").write(text)
        return host.HostScene.from_file(path, RES).to_data()


def models():
    return sorted(os.path.basename(f) for f in glob.glob(RES + "cornell-box/*.obj"))


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    for name in models():
        scene = load_model(name)
        scene.save(os.path.join(OUT, name[:-4] + ".npz"))
        print(name, len(scene.triangles), "triangles", len(scene.nodes), "nodes", len(scene.lights), "lights")
