"""Single-pair inference with reference-format weights.

Mirrors deploy/deploy.py:15-32 (``disp_predict``) and :49-73 (the command line), the checkpoint
wire format of stereo.py:57-92 / utils/utils.py:31-53 (``{'epoch','best_prec','state_dict',
'optim'}`` saved with ``torch.save``; weights-only files ``{'state_dict': ...}``), and the
normalisation transform of myTransforms (``__init__.py:8,121-124``, ``aug_color.py:28-45``).

    python -m dsmnet_amd.deploy --net psmnet --path_weight weight_best.pkl \\
        --path_left 10L.png --path_right 10R.png

Checkpoints are read with ``torch.load(..., weights_only=True)`` only: nothing in the file is
executed.  A file the safe loader refuses is reported, never unpickled.
"""
import argparse
import os
import shutil

import numpy as np
import torch

IMAGENET_MEAN = [0.485, 0.456, 0.406]
IMAGENET_STD = [0.229, 0.224, 0.225]


class Normalize(object):
    """myTransforms/aug_color.py:28-45, in place, on a (3*g, H, W) tensor.

    ``group = min(self.group, img.shape[0] // 3)`` -- so a batched (1,3,H,W) tensor, which is
    what ``disp_predict`` passes, has ``1 // 3 == 0`` groups and comes back UNnormalised.  That
    is the reference's behaviour, weights trained and deployed with it expect it, and it is
    kept."""

    def __init__(self, mean, std, group=1):
        assert len(mean) == 3 and len(std) == 3
        self.group, self.mean, self.std = group, mean, std

    def __call__(self, img):
        group = min(self.group, img.shape[0] // 3)
        for grp in range(group):
            idx = grp * 3
            for i in range(3):
                if self.mean[i] != 0:
                    img[idx + i] -= self.mean[i]
                if self.std[i] != 1:
                    img[idx + i] /= self.std[i]
        return img


def Stereo_normalize():
    """myTransforms/__init__.py:121-124: ImageNet normalisation of a 6-channel L|R stack."""
    return Normalize(IMAGENET_MEAN, IMAGENET_STD, group=2)


# ----------------------------------------------------------------------------
# checkpoint wire format
# ----------------------------------------------------------------------------
def _safe_load(path):
    try:
        return torch.load(path, map_location="cpu", weights_only=True)
    except Exception as first:
        try:                                    # Python-2 pickles: byte strings are latin-1
            return torch.load(path, map_location="cpu", weights_only=True, encoding="latin1")
        except Exception:
            raise RuntimeError("%s was refused by the weights-only loader (%s); it is not "
                               "unpickled any other way" % (path, first))


def load_state(path):
    """The whole checkpoint dict (``state_dict`` plus, for training checkpoints, ``epoch``,
    ``best_prec``, ``optim``)."""
    state = _safe_load(path)
    if not isinstance(state, dict) or "state_dict" not in state:
        raise ValueError("%s is not a reference-format checkpoint: no 'state_dict' entry" % path)
    return state


def load_weights(model, path, strict=True):
    """``model.load_state_dict(torch.load(path)['state_dict'])`` (stereo.py:61-62,
    deploy.py:51-52).  A ``module.`` prefix left by ``nn.DataParallel`` is stripped."""
    sd = load_state(path)["state_dict"]
    if sd and all(k.startswith("module.") for k in sd):
        sd = {k[len("module."):]: v for k, v in sd.items()}
    model.load_state_dict(sd, strict=strict)
    return model


def save_checkpoint(state, is_best, dirpath="./output", filename="model_checkpoint.pkl"):
    """utils/utils.py:31-43: write to ``<file>.tmp``, move into place, copy to
    ``model_best.pkl`` when ``is_best``."""
    if not os.path.exists(dirpath):
        os.makedirs(dirpath)
    path = os.path.join(dirpath, filename)
    torch.save(state, path + ".tmp")
    shutil.move(path + ".tmp", path)
    if is_best:
        shutil.copyfile(path, os.path.join(dirpath, "model_best.pkl"))
    return path


def load_checkpoint(dirpath="./output", best=False):
    """utils/utils.py:46-53: ``None`` when the file does not exist."""
    path = os.path.join(dirpath, "model_best.pkl" if best else "model_checkpoint.pkl")
    if not os.path.exists(path):
        return None
    return load_state(path)


# ----------------------------------------------------------------------------
# prediction
# ----------------------------------------------------------------------------
def disp_predict(model, imgL_np, imgR_np, use_cuda=None):
    """deploy/deploy.py:15-32: (H,W,3) uint8/float RGB arrays -> (H,W) float32 disparity, the
    first of the model's outputs.  ``use_cuda=None``: wherever the model's parameters live."""
    if use_cuda is None:
        use_cuda = next(model.parameters()).is_cuda
    imgL = torch.from_numpy(imgL_np.copy().transpose(2, 0, 1)[None]).float()
    imgR = torch.from_numpy(imgR_np.copy().transpose(2, 0, 1)[None]).float()
    if use_cuda:
        imgL, imgR = imgL.cuda(), imgR.cuda()
    transform = Stereo_normalize()
    imgL = transform(imgL / 255.0)
    imgR = transform(imgR / 255.0)
    with torch.no_grad():
        _, disps = model(imgL, imgR, mode="test")
    d = disps[0]            # (B,1,H,W) for DispNetC / iResNet / GCNet; PSMNet returns (B,H,W), where the
    d = d[0, 0] if d.dim() == 4 else d[0]      # reference's `disps[0][0, 0]` would keep one row only
    return d.detach().cpu().numpy()


def main(argv=None):
    from .img_rw import imread
    from .models import model_create_by_name
    ap = argparse.ArgumentParser(description="stereo matching, single pair (MI355X)")
    ap.add_argument("--net", default="dispnetcorr", type=str,
                    help="dispnetcorr / iresnet / gcnet / psmnet")
    ap.add_argument("--maxdisparity", default=192, type=int)
    ap.add_argument("--path_weight", default="", type=str)
    ap.add_argument("--path_left", default="10L.png", type=str)
    ap.add_argument("--path_right", default="10R.png", type=str)
    ap.add_argument("--flip", default=False, type=bool,
                    help="predict the right view's disparity (mirror both images, swap them)")
    ap.add_argument("--out", default=None, type=str, help="default dispL.png / dispR.png")
    args = ap.parse_args(argv)
    if not torch.cuda.is_available():
        raise SystemExit("dsmnet_amd.deploy needs an MI355X: the HIP path has no CPU fallback")
    imgL, imgR = imread(args.path_left), imread(args.path_right)
    model = model_create_by_name(args.net, args.maxdisparity)
    load_weights(model, args.path_weight)
    model = model.eval().cuda()
    if args.flip:
        disp = disp_predict(model, np.flip(imgR, axis=1), np.flip(imgL, axis=1), True)
        disp, out = np.flip(disp, axis=-1), args.out or "dispR.png"
    else:
        disp, out = disp_predict(model, imgL, imgR, True), args.out or "dispL.png"
    if out.endswith(".pfm"):
        from .img_rw import save_pfm
        save_pfm(out, np.ascontiguousarray(disp, dtype=np.float32))
    else:
        import matplotlib.pyplot as plt
        plt.imsave(out, disp)
    return out


if __name__ == "__main__":
    main()
