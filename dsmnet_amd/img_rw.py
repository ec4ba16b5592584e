"""Image / disparity file IO with the reference's names (myDatasets_stereo/img_rw.py:13-40,
img_rw_pfm.py:13-79): ``imread`` (RGB uint8 or PFM float32), ``imwrite``, ``load_disp``,
``load_pfm``, ``save_pfm``.  The reference decodes with cv2; this image has PIL/matplotlib
instead -- both give the same uint8 RGB array for 8-bit PNG/JPEG.  PFM is written and read in
binary mode (the reference opens it in Py2 text mode)."""
import os
import re

import numpy as np


def load_pfm(fname):
    """-> (image float32, bottom-up rows flipped to top-down; scale).  'PF' = (H,W,3), 'Pf' = (H,W)."""
    if not os.path.isfile(fname):
        raise IOError("no such PFM file: %s" % fname)
    with open(fname, "rb") as f:
        header = f.readline().rstrip()
        if header == b"PF":
            color = True
        elif header == b"Pf":
            color = False
        else:
            raise ValueError("Not a PFM file.")
        m = re.match(br"^(\d+)\s(\d+)\s$", f.readline())
        if not m:
            raise ValueError("Malformed PFM header.")
        width, height = int(m.group(1)), int(m.group(2))
        scale = float(f.readline().rstrip())
        endian = "<" if scale < 0 else ">"
        scale = abs(scale)
        data = np.frombuffer(f.read(), dtype=endian + "f4")
    shape = (height, width, 3) if color else (height, width)
    if data.size != int(np.prod(shape)):
        raise ValueError("PFM payload has %d floats, header says %s" % (data.size, shape))
    return np.flipud(data.reshape(shape)).astype(np.float32).copy(), scale


def save_pfm(fname, image, scale=1):
    if image.dtype.name != "float32":
        raise ValueError("Image dtype must be float32.")
    if image.ndim == 3 and image.shape[2] == 3:
        color = True
    elif image.ndim == 2 or (image.ndim == 3 and image.shape[2] == 1):
        color = False
    else:
        raise ValueError("Image must have H x W x 3, H x W x 1 or H x W dimensions.")
    image = np.flipud(image)
    little = image.dtype.byteorder == "<" or (image.dtype.byteorder == "=" and np.little_endian)
    with open(fname, "wb") as f:
        f.write(b"PF\n" if color else b"Pf\n")
        f.write(("%d %d\n" % (image.shape[1], image.shape[0])).encode())
        f.write(("%f\n" % (-scale if little else scale)).encode())
        f.write(np.ascontiguousarray(image).tobytes())


def _decode(fname):
    try:
        from PIL import Image
        with Image.open(fname) as im:
            return np.array(im.convert("RGB") if im.mode not in ("L", "I;16", "I", "F") else im)
    except ImportError:
        import matplotlib.pyplot as plt
        a = plt.imread(fname)
        if a.dtype != np.uint8:
            a = np.round(a * 255.0).astype(np.uint8)
        return a[..., :3] if a.ndim == 3 else a


def imread(fname):
    """RGB uint8 (H,W,3) for ordinary images, float32 for .pfm (img_rw.py:23-29)."""
    if fname.find(".pfm") > 0:
        return load_pfm(fname)[0]
    return np.array(_decode(fname))


def imwrite(fname, image):
    if fname.find(".pfm") > 0:
        return save_pfm(fname, image)
    try:
        from PIL import Image
        Image.fromarray(np.ascontiguousarray(image)).save(fname)
    except ImportError:
        import matplotlib.pyplot as plt
        plt.imsave(fname, image)


def load_gray(fname):
    gray = imread(fname)
    if gray.ndim > 2:
        gray = gray[:, :, 0]
    gray = np.array(gray)
    if gray.dtype.kind == "f":
        gray[~np.isfinite(gray)] = 0       # the reference zeroes inf (its `== nan` test is a no-op)
    return gray


def load_disp(fname):
    return load_gray(fname)
