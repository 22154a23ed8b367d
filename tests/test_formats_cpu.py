"""CPU tests of the data formats either side of the path: the reference's PLY point-cloud schema,
deformation.pth, and the numpy restatement of compute_3D_filter (known answers)."""
import math
import os
import sys
from types import SimpleNamespace

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "e-d3dgs_amd"))
sys.path.insert(0, ROOT)


def _model(P=37):
    from ed3dgs_amd import synthetic as S
    from ed3dgs_amd.model import SynthGaussianModel
    scene = S.make_scene(P, seed=9)
    scene.tongue_class = (torch.rand(P, 1) > 0.5).float()
    scene.filter_3D = torch.rand(P, 1) * 0.01
    return SynthGaussianModel(scene, device="cpu")


def test_ply_schema_matches_reference_attribute_list(tmp_path):
    """Header = what plyfile writes for the reference's dtype_full (scene/gaussian_model.py:231-248, :275-283)."""
    m = _model()
    path = str(tmp_path / "point_cloud" / "iteration_7" / "point_cloud.ply")
    m.save_ply(path)
    raw = open(path, "rb").read()
    head, _, body = raw.partition(b"end_header\n")
    lines = head.decode("ascii").strip().split("\n")
    assert lines[:3] == ["ply", "format binary_little_endian 1.0", "element vertex 37"]
    expect = (["x", "y", "z", "nx", "ny", "nz"] + ["f_dc_%d" % i for i in range(3)] + ["f_rest_%d" % i for i in range(45)] +
              ["opacity"] + ["scale_%d" % i for i in range(3)] + ["rot_%d" % i for i in range(4)] +
              ["embedding_%d" % i for i in range(32)] + ["tongue_class", "filter_3D"])
    assert lines[3:] == ["property float %s" % n for n in expect]
    assert len(body) == 37 * 4 * len(expect)
    first = np.frombuffer(body[:12], "<f4")
    np.testing.assert_array_equal(first, m._xyz.detach().numpy()[0])
    # f_dc / f_rest are written channel-major (transpose(1, 2).flatten, :266-267)
    off = 4 * (6 + 3)
    rest0 = np.frombuffer(body[off:off + 4 * 45], "<f4")
    np.testing.assert_array_equal(rest0, m._features_rest.detach().transpose(1, 2).flatten(start_dim=1).numpy()[0])


def test_ply_round_trip_and_missing_tongue_class(tmp_path):
    from ed3dgs_amd import ply
    m = _model()
    path = str(tmp_path / "pc.ply")
    m.save_ply(path)
    m2 = _model(5)
    m2.load_ply(path, device="cpu")
    for n in ("_xyz", "_opacity", "_scaling", "_rotation", "_embedding", "_features_dc", "_features_rest", "filter_3D", "tongue_class"):
        a, b = getattr(m, n).detach(), getattr(m2, n).detach()
        assert a.shape == b.shape, n
        assert torch.equal(a, b), n
    assert m2._features_rest.shape == (37, 15, 3) and m2._features_dc.shape == (37, 1, 3)
    # a file without tongue_class (older checkpoints): zeros (:303-306); ascii variant of the same data
    v = ply.read_vertices(path)
    keep = [n for n in v.dtype.names if n != "tongue_class"]
    sub = np.empty(len(v), dtype=[(n, "f4") for n in keep])
    for n in keep:
        sub[n] = v[n]
    p2 = str(tmp_path / "no_tongue.ply")
    with open(p2, "w") as f:
        f.write("ply\nformat ascii 1.0\ncomment written by a test\nelement vertex %d\n" % len(sub))
        for n in keep:
            f.write("property float %s\n" % n)
        f.write("end_header\n")
        for r in sub:
            f.write(" ".join(repr(float(x)) for x in r) + "\n")
    m3 = _model(5)
    m3.load_ply(p2, device="cpu")
    assert torch.equal(m3._xyz.detach(), m._xyz.detach()) and float(m3.tongue_class.abs().max()) == 0.0


def test_deformation_pth_round_trip(tmp_path):
    m = _model()
    with torch.no_grad():
        for p in m._deformation.parameters():
            p.add_(torch.randn_like(p) * 0.01)
    m.save_deformation(str(tmp_path))
    m2 = _model()
    m2.load_model(str(tmp_path), device="cpu")
    for (n1, a), (n2, b) in zip(m._deformation.state_dict().items(), m2._deformation.state_dict().items()):
        assert n1 == n2 and torch.equal(a, b), n1


def _cam(R=np.eye(3), T=np.zeros(3), W=100, H=100, fov=math.pi / 2):
    return SimpleNamespace(R=R, T=T, FoVx=fov, FoVy=fov, image_width=W, image_height=H)


def test_compute_3d_filter_known_answers():
    """One camera at the origin looking down +z, 100x100, fov 90 deg -> focal 50 (scene/gaussian_model.py:538-592)."""
    from oracle import filter3d_ref as F
    s = 0.2 ** 0.5
    xyz = np.array([[0, 0, 2.0],        # seen: filter = 2 / 50 * sqrt(0.2)
                    [0, 0, -1.0],       # behind: takes the largest seen distance (4)
                    [0, 0, 0.1],        # z <= 0.2: not seen
                    [4.0, 0, 4.0],      # x/z*50 + 50 = 100 <= 115: seen
                    [6.0, 0, 4.0],      # 125 > 115: not seen
                    [-5.1, 0, 4.0]],    # -13.75 >= -15: seen
                   np.float32)
    f = F.compute_3D_filter(xyz, [_cam()])[:, 0]
    np.testing.assert_allclose(f, np.array([2, 4, 4, 4, 4, 4], np.float32) / 50 * s, rtol=1e-6)
    # second camera closer to point 0 (translated by +1.5 along z: z_cam = z - ... uses xyz @ R + T), larger focal wins
    f2 = F.compute_3D_filter(xyz, [_cam(), _cam(T=np.array([0, 0, -1.0]), W=200, H=200)])[:, 0]
    assert abs(f2[0] - 1.0 / 100 * s) < 1e-7            # min z = 1, focal = max(50, 100)
    assert F.compute_3D_filter(xyz[1:3], [_cam()]).max() == 0.0   # nothing seen at all
    rows = F.camera_rows([_cam(W=200, H=100)])
    assert rows.shape == (1, 16) and abs(rows[0, 12] - 100.0) < 1e-4 and abs(rows[0, 13] - 50.0) < 1e-4


def test_cfg_args_is_parsed_without_eval(tmp_path):
    """arguments/__init__.py:174-194 evals the file; here only literals are accepted."""
    from argparse import Namespace
    from ed3dgs_amd.model import read_cfg_args, write_cfg_args
    ns = Namespace(sh_degree=3, net_width=64, source_path="/data/x y", white_background=False, lr=1.6e-4, cams=[1, 2], none=None)
    write_cfg_args(str(tmp_path), ns)
    assert open(tmp_path / "cfg_args").read() == str(ns)          # train.py:483-484
    got = read_cfg_args(str(tmp_path))
    assert vars(got) == vars(ns)
    assert vars(read_cfg_args(str(tmp_path / "missing"))) == {}
    for bad in ("Namespace(a=__import__('os').system('true'))", "print(1)", "Namespace(1)", "Namespace(**{})"):
        (tmp_path / "cfg_args").write_text(bad)
        with pytest.raises(ValueError):
            read_cfg_args(str(tmp_path))


def test_ply_bytes_equal_what_the_references_save_ply_hands_to_plyfile(tmp_path):
    """tests/golden/ply_reference.npz (tools/gen_densify_golden.py): the REFERENCE's own save_ply (scene/gaussian_model.py:261-283) run on
    a 37-Gaussian model with capturing stand-ins for plyfile's classes -- the attribute names in construct_list_of_attributes order and
    the structured vertex array's bytes (what plyfile writes behind the header for an all-f4 binary_little_endian element).
    SynthGaussianModel.save_ply must produce the same property list and the same body, byte for byte."""
    from ed3dgs_amd import synthetic as S
    from ed3dgs_amd.model import SynthGaussianModel
    z = np.load(os.path.join(ROOT, "tests", "golden", "ply_reference.npz"))
    scene = S.make_scene(37, seed=9)
    for k, attr in (("xyz", "xyz"), ("f_dc", "f_dc"), ("f_rest", "f_rest"), ("opacity", "opacity"), ("log_scale", "log_scale"),
                    ("rot", "rot"), ("embedding", "embedding")):
        np.testing.assert_array_equal(getattr(scene, attr).numpy(), z[k])        # same synthetic model as the generator's
    scene.tongue_class = torch.from_numpy(z["tongue_class"])
    scene.filter_3D = torch.from_numpy(z["filter_3D"])
    m = SynthGaussianModel(scene, device="cpu")
    path = str(tmp_path / "point_cloud" / "iteration_7" / "point_cloud.ply")
    m.save_ply(path)
    head, _, body = open(path, "rb").read().partition(b"end_header\n")
    lines = head.decode("ascii").strip().split("\n")
    assert lines[3:] == ["property float %s" % n for n in z["names"].tolist()]
    assert body == z["vertex_bytes"].tobytes()
