"""Mirror of lib/datasets/dataset.py (`Dataset`) and lib/datasets/dataloader.py (`collate_fn`) for the geometry side
of a sample: same constructor, `__len__`, `__getitem__` keys ('id','data','gt_box','gt_mask','nearest_images',
'image_files'), modes and filters.  Differences:

* parsing is zero-copy (datasets/scene_file.py) and, with `device_encode=True`, the 2-channel TSDF input is built on
  the GPU straight from the file's sdf block (`ops.tsdf_encode`) in the conv stack's channels-last layout -- 'data' is
  then a CUDA tensor of logical shape (2,X,Y',Z); with `device_encode=False` it is the reference's numpy array;
* colour frames (USE_IMAGES, dataset.py:136-190): depth PNG, colour / label image and pose file of every frame the
  container (chunk mode) or the scene's `depth/` directory (scene / benchmark mode) names are decoded with Pillow -- the
  library behind the reference's `scipy.misc.imread` and `torchvision.transforms` -- and resized / cropped / normalised by
  the same rules (`frames.py`); `collate_fn` stacks them into blobs['nearest_images'] exactly as dataloader.py:19-40.
"""
import csv
import math
import os

import numpy as np
import torch

from . import frames
from .scene_file import SceneFile


class Dataset(torch.utils.data.Dataset):
    def __init__(self, data_location, mode, cfg, device_encode=False):
        super().__init__()
        self.mode = mode                      # chunk | scene | benchmark (dataset.py:24-29)
        self.cfg = cfg
        self.device_encode = device_encode
        if os.path.isdir(data_location):
            self.scenes = [os.path.join(data_location, x) for x in os.listdir(data_location)
                           if os.path.isfile(os.path.join(data_location, x))]
        elif os.path.isfile(data_location):
            with open(data_location) as f:
                self.scenes = [x.strip() for x in f.readlines()]
        else:
            raise FileNotFoundError(data_location)
        if cfg.LABEL_MAP != "":
            self.mapping, self.weights = Dataset.load_mapping(cfg.LABEL_MAP)

    def __len__(self):
        return len(self.scenes)

    @staticmethod
    def load_mapping(label_file):
        """dataset.py:269-283: nyu40id -> consecutive id, class weights with the background weight first"""
        mapping, by_id = {}, {}
        with open(label_file) as f:
            for row in csv.DictReader(f, delimiter=","):
                mapping[int(row["nyu40id"])] = int(row["mappedIdConsecutive"])
                by_id[int(row["mappedIdConsecutive"])] = float(row["weight"])
        return mapping, [0.3280746813009404] + [by_id[k] for k in sorted(by_id)]

    @staticmethod
    def outbbox_thresh(b):
        """dataset.py:220-229: fraction of a box inside the 96x48x96 chunk"""
        lim = (96, 48, 96)
        lo = [min(max(b[k], 0), lim[k]) for k in range(3)]
        hi = [min(max(b[3 + k], 0), lim[k]) for k in range(3)]
        return ((hi[0] - lo[0]) * (hi[1] - lo[1]) * (hi[2] - lo[2])) / ((b[3] - b[0]) * (b[4] - b[1]) * (b[5] - b[2]))

    def encode_tsdf(self, sf, max_height):
        cfg = self.cfg
        mode = "flip" if cfg.get("FLIP_TSDF", False) else ("log" if cfg.get("LOG_TSDF", False) else "abs")
        if self.device_encode:
            from .. import ops
            raw = torch.from_numpy(np.array(sf.sdf)).cuda()        # blocking: the staging copy is pageable (ROCm hazards)
            # planar (2,X,Y,Z), the layout the reference's blobs['data'] has and the first-layer kernels read
            return ops.tsdf_encode(raw, sf.dims, cfg.TRUNCATED, mode, max_height, channels_last=False)[0]
        v = sf.sdf_grid()[None].astype(np.float32)
        a = np.abs(np.clip(v, -cfg.TRUNCATED, cfg.TRUNCATED))
        if mode == "flip":
            a = cfg.TRUNCATED - a
        elif mode == "log":
            a = np.log(a)
        return np.concatenate([a, np.greater(v, -1)], 0)[:, :, :max_height, :]

    def __getitem__(self, idx):
        cfg = self.cfg
        need_masks = bool(cfg.USE_MASK or cfg.KEEP_THRESH or cfg.USE_IMAGES)
        need_stats = bool(cfg.KEEP_THRESH or cfg.USE_IMAGES)
        sf = SceneFile(self.scenes[idx], want_masks=need_masks, want_stats=need_stats, want_images=bool(cfg.USE_IMAGES))
        max_height = 480 if self.mode == "benchmark" else 48
        data = self.encode_tsdf(sf, max_height)
        gt_box = np.zeros((len(sf.boxes), 7), dtype=np.float32)
        for i, (b, lab) in enumerate(zip(sf.boxes, sf.box_labels)):
            lab = int(lab)
            if cfg.LABEL_MAP != "":
                lab = self.mapping[lab]
            gt_box[i] = [math.floor(b[0]), math.floor(b[1]), math.floor(b[2]), math.ceil(b[3]), math.ceil(b[4]), math.ceil(b[5]), lab]
        gt_mask = []
        for _, m in sf.masks:
            m = m.astype(np.uint8)                      # the reference's uint16 -> uint8 wrap, then everything > 1 cleared
            m[m > 1] = 0
            gt_mask.append(m)
        if need_stats:
            stats = [self.outbbox_thresh(gt_box[i]) if self.mode == "chunk" else float(p)
                     for i, p in enumerate(sf.part_in_volume)]
            keep = [i for i, p in enumerate(stats) if p >= cfg.KEEP_THRESH and self.weights[int(gt_box[i, 6])] != 0]
            gt_box = gt_box[keep]
            if cfg.USE_MASK:
                gt_mask = [gt_mask[i] for i in keep]
        nearest_images, image_files = {}, []
        if cfg.USE_IMAGES:
            nearest_images, image_files = self.load_frames(idx, sf)
        boxes, masks = [], []
        for i, b in enumerate(gt_box):
            if b[1] <= max_height and b[4] <= max_height:
                boxes.append(b)
                masks.append(gt_mask[i])
        return {"id": self.scenes[idx], "data": data, "gt_box": np.array(boxes), "gt_mask": masks,
                "nearest_images": nearest_images, "image_files": image_files}

    def scene_name(self, idx):
        """dataset.py:146-151: frame directory of a sample, from the layout BASE_IMAGE_PATH ends with"""
        base = self.cfg.BASE_IMAGE_PATH.rstrip("/")
        name = os.path.basename(self.scenes[idx])
        if base.endswith("augmented"):
            return name.rsplit("_", 1)[0] if self.mode == "chunk" else name.split(".")[0]
        if base.endswith("square"):
            return name.split("__")[0]
        raise NotImplementedError("BASE_IMAGE_PATH must end with 'augmented' or 'square' (dataset.py:146-151)")

    def load_frames(self, idx, sf):
        """dataset.py:136-190 -> (nearest_images dict, image_files): chunk mode reads the container's frame ids and its
        world2chunk; scene / benchmark mode takes every file of <scene>/depth/ (os.listdir order, as the reference) and
        the scene's world2grid.txt with the (10,16,10) padding subtracted."""
        cfg = self.cfg
        root = os.path.join(cfg.BASE_IMAGE_PATH, self.scene_name(idx))
        if self.mode == "chunk":
            world2grid = np.linalg.inv(np.transpose(sf.world2chunk).astype(np.float32))
            ids = [int(v) for v in sf.frame_ids]
        else:
            world2grid = frames.load_pose(os.path.join(root, "world2grid.txt"))
            world2grid[0][3] -= 10
            world2grid[1][3] -= 16
            world2grid[2][3] -= 10
            ids = [f.split(".")[0] for f in os.listdir(os.path.join(root, "depth"))]
        relabel = bool(cfg.USE_IMAGES_GT and cfg.LABEL_MAP != "")
        depths, images, poses, files = [], [], [], []
        for fid in ids:
            image_file = os.path.join(root, cfg.IMAGE_TYPE, str(fid) + cfg.IMAGE_EXT)
            poses.append(frames.load_pose(os.path.join(root, "pose", str(fid) + ".txt")))
            depths.append(frames.load_depth(os.path.join(root, "depth", str(fid) + ".png"), cfg.DEPTH_SHAPE))
            im = frames.load_image(image_file, cfg.IMAGE_SHAPE, cfg.COLOR_MEAN, cfg.COLOR_STD)
            if relabel:
                im = frames.relabel(im, self.mapping, self.weights)
            images.append(im)
            files.append(image_file)
        return {"depths": depths, "images": images, "poses": poses, "world2grid": world2grid, "frameids": ids}, files


def collate_images(batch, cfg):
    """dataloader.py:16-40: per sample the (V,...) stacks of images / depths / poses and world2grid expanded to V views;
    in train mode the view list is cut to NUM_IMAGES (or a random count in [1, NUM_IMAGES] with RANDOM_NUM_IMAGES)"""
    depths, poses, world2grid = [], [], []
    for b in batch:
        x = b["nearest_images"]
        n = len(x["depths"])
        cap = cfg.NUM_IMAGES if not cfg.get("RANDOM_NUM_IMAGES", False) else np.random.randint(low=1, high=cfg.NUM_IMAGES + 1)
        if cap < n and cfg.get("MODE", "benchmark") == "train":
            n = cap
            x["images"], x["depths"], x["poses"] = x["images"][:n], x["depths"][:n], x["poses"][:n]
        depths.append(torch.from_numpy(np.array(x["depths"])))
        poses.append(torch.from_numpy(np.array(x["poses"])))
        world2grid.append(torch.from_numpy(x["world2grid"]).expand(n, 4, 4))
    images = [torch.from_numpy(np.stack([np.asarray(i) for i in b["nearest_images"]["images"]], 0).astype(np.float32)) for b in batch]
    return {"images": images, "depths": depths, "poses": poses, "world2grid": world2grid}


def collate_fn(batch, cfg=None):
    """dataloader.py:8-49 (batch of dicts -> blobs for Network.forward).  The reference reads its module-global cfg; here
    the colour side is collated when the samples carry frames (or `cfg.USE_IMAGES` is given)."""
    def tens(x):
        return x if torch.is_tensor(x) else torch.from_numpy(x)
    with_images = bool(batch[0]["nearest_images"]) and "depths" in batch[0]["nearest_images"]
    if cfg is None:
        from ..config import cfg as cfg_default
        cfg = cfg_default
    nearest = collate_images(batch, cfg) if with_images else {}
    if len(batch) == 1 and torch.is_tensor(batch[0]["data"]):
        data = batch[0]["data"].unsqueeze(0)            # device-encoded sample: stays on the GPU, planar like the host path
    else:
        data = torch.stack([tens(x["data"]) for x in batch], 0)
    return {"id": [x["id"] for x in batch],
            "data": data,
            "gt_box": [torch.from_numpy(x["gt_box"]) for x in batch if x["gt_box"].shape[0] != 0],
            "gt_mask": [[torch.from_numpy(y) for y in x["gt_mask"]] for x in batch if len(x["gt_mask"]) != 0],
            "nearest_images": nearest, "image_files": batch[0]["image_files"]}


def get_dataloader(dataset, batch_size=1, shuffle=False, num_workers=0):
    import functools
    return torch.utils.data.DataLoader(dataset, batch_size=batch_size, collate_fn=functools.partial(collate_fn, cfg=dataset.cfg),
                                       shuffle=shuffle, num_workers=num_workers)
