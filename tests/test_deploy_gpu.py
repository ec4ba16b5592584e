"""deploy path on the GPU: reference-format weights file -> model ->
``disp_predict`` on a synthetic uint8 pair, against the oracle run on the same weights."""
import numpy as np
import pytest
import torch

from oracle import models as OM
from tests.golden.make_goldens import randomise_bn

pytestmark = pytest.mark.gpu


def _pair(h, w, seed):
    rng = np.random.default_rng(seed)
    left = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    return left, np.roll(left, -5, axis=1)


def test_disp_predict_from_weight_file(hip_lib, tmp_path, net="dispnetcorr", size=(256, 512), tol=1e-3):
    from dsmnet_amd import deploy
    from dsmnet_amd.models import model_create_by_name
    sd = randomise_bn(OM.init_state(net, 3), 9)
    path = str(tmp_path / "weight_best.pkl")
    torch.save({"state_dict": sd}, path)                       # stereo.py:81-83
    model = deploy.load_weights(model_create_by_name(net, 192), path).eval().cuda()
    imgL, imgR = _pair(size[0], size[1], 1)
    got = deploy.disp_predict(model, imgL, imgR)
    assert got.shape == size and got.dtype == np.float32
    # oracle on the same tensors: img / 255, no normalisation (Normalize sees 1 // 3 == 0 groups)
    tl = torch.from_numpy(imgL.transpose(2, 0, 1)[None].copy()).float() / 255.0
    tr = torch.from_numpy(imgR.transpose(2, 0, 1)[None].copy()).float() / 255.0
    with torch.no_grad():
        want = OM.forward(net, sd, tl, tr)[1][0][0, 0].numpy()
    assert np.abs(got - want).max() <= tol
    # --flip: right-view disparity = mirrored prediction of the mirrored, swapped pair
    flipped = deploy.disp_predict(model, np.flip(imgR, axis=1), np.flip(imgL, axis=1))
    assert flipped.shape == size


def test_disp_predict_psmnet_returns_a_full_map(hip_lib):
    """PSMNet's outputs are (B,H,W): `disps[0][0, 0]` of the reference would keep one row."""
    from dsmnet_amd import deploy
    from dsmnet_amd.models import model_create_by_name
    torch.manual_seed(0)
    model = model_create_by_name("psmnet", 192).eval().cuda()
    imgL, imgR = _pair(256, 512, 4)
    d = deploy.disp_predict(model, imgL, imgR)
    assert d.shape == (256, 512) and np.isfinite(d).all()


def test_cli_writes_a_disparity_map(hip_lib, tmp_path):
    from dsmnet_amd import deploy, img_rw
    sd = OM.init_state("dispnetcorr", 0)
    w = str(tmp_path / "w.pkl")
    torch.save({"epoch": 3, "best_prec": 2.0, "state_dict": sd, "optim": {}}, w)
    imgL, imgR = _pair(256, 512, 2)
    pl, pr, out = str(tmp_path / "L.png"), str(tmp_path / "R.png"), str(tmp_path / "d.pfm")
    img_rw.imwrite(pl, imgL)
    img_rw.imwrite(pr, imgR)
    assert deploy.main(["--net", "dispnetcorr", "--path_weight", w, "--path_left", pl,
                        "--path_right", pr, "--out", out]) == out
    d = img_rw.load_disp(out)
    assert d.shape == (256, 512) and np.isfinite(d).all()
