"""Per-frame files of the colour path (lib/datasets/dataset.py:230-267): camera pose text, depth PNG, colour / label image.

The reference decodes with `scipy.misc.imread` (scipy 1.1: `PIL.Image.open` + `numpy.array`) and resizes with
`torchvision.transforms.Resize(..., interpolation=Image.NEAREST)` / `CenterCrop` (both thin wrappers over PIL) and
`transforms.Normalize`.  Neither scipy.misc nor torchvision exists in this image; Pillow -- the library that does the work
underneath both -- does, so the same three steps are written against it directly:

    Resize([h, w], NEAREST)      == img.resize((w, h), Image.NEAREST)                      (torchvision functional.resize)
    CenterCrop([h, w])           == img.crop((left, top, left + w, top + h)) with
                                    top = int(round((H - h) / 2.)), left = int(round((W - w) / 2.))   (functional.center_crop)
    Normalize(mean, std)(t)      == (t - mean[:, None, None]) / std[:, None, None]
"""
import math

import numpy as np
import torch


def _image_module():
    try:
        from PIL import Image
    except ImportError as e:                                  # pragma: no cover
        raise RuntimeError("decoding frame images needs Pillow (the reference's own dependency)") from e
    return Image


def load_pose(filename):
    """dataset.py:230-235: four lines of four blank-separated numbers -> float32 (4,4)"""
    with open(filename) as f:
        lines = f.read().splitlines()
    assert len(lines) == 4
    rows = [x.split(" ")[:4] for x in lines]
    return np.asarray(rows).astype(np.float32)


def imread(filename):
    """scipy.misc.imread(name) of scipy <= 1.1 (flatten=False, mode=None): palette images are expanded, bilevel images
    become 8-bit, everything else is `numpy.array(img)` (uint8 (H,W[,3|4]); 16-bit depth PNGs come out as uint16/int32)"""
    Image = _image_module()
    im = Image.open(filename)
    if im.mode == "P":
        im = im.convert("RGBA" if "transparency" in im.info else "RGB")
    elif im.mode == "1":
        im = im.convert("L")
    return np.array(im)


def resize_crop_image(image, new_image_dims):
    """dataset.py:237-246: scale to the target HEIGHT with nearest sampling (width by the aspect ratio, floored), then
    centre-crop the width.  new_image_dims = [width, height]."""
    Image = _image_module()
    image_dims = [image.shape[1], image.shape[0]]
    if image_dims == list(new_image_dims):
        return image
    new_w, new_h = int(new_image_dims[0]), int(new_image_dims[1])
    resize_width = int(math.floor(new_h * float(image_dims[0]) / float(image_dims[1])))
    img = Image.fromarray(image).resize((resize_width, new_h), Image.NEAREST)
    top = int(round((new_h - new_h) / 2.0))
    left = int(round((resize_width - new_w) / 2.0))
    return np.array(img.crop((left, top, left + new_w, top + new_h)))


def load_depth(filename, image_dims):
    """dataset.py:248-253: millimetres -> metres, float32 (h, w)"""
    return resize_crop_image(imread(filename), image_dims).astype(np.float32) / 1000.0


def load_image(filename, image_dims, mean, std):
    """dataset.py:255-267: colour image -> normalised float32 torch tensor (3,h,w); label image -> (1,h,w) array"""
    image = resize_crop_image(imread(filename), image_dims)
    if image.ndim == 3:
        t = torch.from_numpy(np.transpose(image, [2, 0, 1]).astype(np.float32) / 255.0)
        m = torch.as_tensor(mean, dtype=torch.float32).view(-1, 1, 1)
        s = torch.as_tensor(std, dtype=torch.float32).view(-1, 1, 1)
        return (t - m) / s
    if image.ndim == 2:
        return np.expand_dims(image, 0)
    raise ValueError("unsupported image rank %d" % image.ndim)


def relabel(im_pre, mapping, weights):
    """dataset.py:171-178 (USE_IMAGES_GT with a label map): nyu40 ids above 40 cleared, ids mapped to the consecutive
    training ids, zero-weight classes to 0; comparisons are made against the UNMAPPED copy"""
    im_pre = np.where(im_pre <= 40, im_pre, 0)
    im_post = im_pre.copy()
    for k, v in mapping.items():
        if weights[v] == 0:
            v = 0
        im_pre[im_post == k] = v
    return im_pre
