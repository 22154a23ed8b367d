"""GPU tests of SURVEY 8(f) rank 3: compute_3D_filter through the C ABI against the numpy restatement, and a checkpoint
(point_cloud.ply + deformation.pth) written, re-loaded and rendered to the same image."""
import numpy as np
import pytest
import torch

import util  # noqa: F401  (puts the package on sys.path)

pytestmark = pytest.mark.gpu


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


@pytest.mark.parametrize("P,n_cams", [(50_000, 16), (1000, 40), (7, 1)])
def test_compute_3d_filter_matches_oracle(P, n_cams):
    _need_gpu()
    from ed3dgs_amd import synthetic as S
    from ed3dgs_amd.filter3d import compute_3D_filter
    from oracle import filter3d_ref as F
    scene = S.make_scene(P, seed=4)
    xyz = scene.xyz * 3.0            # spread beyond the frusta so that some points are seen by no camera
    cams = S.make_cameras(n_cams, 640, 360, seed=6)
    ref = F.compute_3D_filter(xyz.numpy(), cams)
    got = compute_3D_filter(xyz.cuda(), cams).cpu().numpy()
    assert got.shape == ref.shape
    # same float32 operations in the same order on both sides: equal up to a threshold decision at the last ulp
    close = np.isclose(got, ref, rtol=1e-6, atol=0)
    assert close.mean() > 0.9999, close.mean()
    if P >= 1000:
        assert np.unique(ref).size > P // 4    # the comparison is not vacuous


def test_compute_3d_filter_edge_cases():
    _need_gpu()
    from ed3dgs_amd import synthetic as S
    from ed3dgs_amd.filter3d import compute_3D_filter
    cams = S.make_cameras(2, 64, 64)
    assert compute_3D_filter(torch.zeros(0, 3).cuda(), cams).shape == (0, 1)
    far = torch.full((5, 3), 1000.0).cuda()           # seen by nobody -> zeros (the reference raises on the empty max)
    assert float(compute_3D_filter(far, cams).abs().max()) == 0.0
    assert float(compute_3D_filter(torch.zeros(3, 3).cuda(), []).abs().max()) == 0.0
    with pytest.raises(RuntimeError):
        compute_3D_filter(torch.zeros(3, 3), cams)


def test_checkpoint_round_trip_renders_identically(tmp_path):
    _need_gpu()
    from ed3dgs_amd import synthetic as S
    from ed3dgs_amd.model import PIPE, SynthGaussianModel, load_checkpoint
    from gaussian_renderer import render
    scene = S.make_scene(3000, seed=2)
    m = SynthGaussianModel(scene, device="cuda")
    cams = S.make_cameras(3, 256, 192, device="cuda")
    m.compute_3D_filter(cams)
    assert float(m.filter_3D.min()) > 0
    d = tmp_path / "point_cloud" / "iteration_30000"
    m.save_ply(str(d / "point_cloud.ply"))
    m.save_deformation(str(d))
    m2 = load_checkpoint(str(tmp_path), 30000, args=m.args, device="cuda")
    bg = torch.zeros(3, device="cuda")
    cam = cams[1].with_time(0.4)
    with torch.no_grad():
        kw = dict(iter=20000, num_down_emb_c=30, num_down_emb_f=30, cam_no=0, require_coord=False, require_depth=True,
                  disable_filter3D=False)      # the 3D filter read back from the PLY takes part
        a = render(cam, m, PIPE, bg, 0.1, **kw)
        b = render(cam, m2, PIPE, bg, 0.1, **kw)
    for k in ("render", "mask", "expected_depth", "median_depth", "normal", "radii"):
        assert torch.equal(a[k], b[k]), k
    assert float(a["mask"].max()) > 0.5


def test_image_stats_matches_torch():
    """ed3dgs_image_stats (csrc/stats.hip): {sum(image * weight), -10 log10(mean((image - mid)^2)), 1} in one launch, scratch left
    zero for the next call; odd element counts take the scalar tail."""
    _need_gpu()
    import ctypes as C
    from ed3dgs_amd import _lib
    L = _lib.lib()
    g = torch.Generator(device="cuda").manual_seed(3)
    acc = torch.zeros(272, device="cuda")   # ED3DGS_STATS_ACC_FLOATS
    for n in (3 * 1080 * 1920, 1027, 4, 3):
        img = torch.rand(n, generator=g, device="cuda")
        w = torch.randn(n, generator=g, device="cuda") / n
        out = torch.full((3,), -1.0, device="cuda")
        for _ in range(2):                                   # twice: the first call must leave the scratch clean
            rc = L.ed3dgs_image_stats(C.c_void_p(img.data_ptr()), C.c_void_p(w.data_ptr()), C.c_size_t(n), C.c_float(0.5),
                                      C.c_void_p(acc.data_ptr()), C.c_void_p(out.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream))
            assert rc == 0
            want0 = float(torch.dot(img.double(), w.double()))
            want1 = float(-10.0 * torch.log10((img.double() - 0.5).square().mean()))
            got = out.cpu().tolist()
            assert abs(got[0] - want0) <= 1e-5 * max(1e-3, abs(want0)) + 1e-7 and abs(got[1] - want1) <= 1e-4 and got[2] == 1.0
        assert float(acc.abs().max()) == 0.0
