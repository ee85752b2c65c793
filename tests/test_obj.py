"""OBJ reader (pytorch3d.io.load_obj stand-in for the head template)."""
import os

import numpy as np
import pytest

from fateavatar_amd import scenes
from fateavatar_amd.obj import load_obj


def test_small_obj_with_quads_uvs_and_relative_indices(tmp_path):
    p = tmp_path / "m.obj"
    p.write_text("# comment\nv 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvt 0 0\nvt 1 0\nvt 1 1\nvt 0 1\nvn 0 0 1\n"
                 "f 1/1/1 2/2/1 3/3/1 4/4/1\nf -4//1 -3//1 -2//1\nf 1 3 4\n")
    m = load_obj(str(p))
    assert m["verts"].shape == (4, 3) and m["verts_uvs"].shape == (4, 2)
    assert m["faces"].tolist() == [[0, 1, 2], [0, 2, 3], [0, 1, 2], [0, 2, 3]]
    assert m["faces_uvs"].tolist() == [[0, 1, 2], [0, 2, 3], [-1, -1, -1], [-1, -1, -1]]


@pytest.mark.skipif(not os.path.exists("/root/reference/weights/head_template_mouth_close.obj"),
                    reason="the reference checkout is only present in the build container")
def test_head_template_matches_the_committed_geometry_fixture():
    m = load_obj("/root/reference/weights/head_template_mouth_close.obj")
    verts, faces, kind = scenes.head_geometry()
    assert kind == "head_template"
    assert np.array_equal(m["verts"], verts) and np.array_equal(m["faces"], faces)
    assert m["verts_uvs"].shape[1] == 2 and m["faces_uvs"].shape == faces.shape and m["faces_uvs"].min() >= 0
