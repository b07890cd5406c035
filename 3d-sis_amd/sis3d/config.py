"""Forward-path configuration.

Mirrors the forward-relevant keys of the reference's module-global ``cfg``
(lib/utils/config.py:12-247) with the values of
experiments/cfgs/ScanNet/benchmark.yml as defaults.  Attribute names are the
reference's own so that code reading ``cfg.X`` looks the same on both sides.
``sis3d.dropin`` can also adopt the reference's live ``cfg`` object.
"""
import copy

# experiments/anchors/*.txt (size_x, size_y, size_z per anchor, scene voxels)
ANCHOR_SETS = {
    "scannet14_3.txt": [(8, 9, 8), (14, 11, 14), (14, 20, 14)],
    "scannet14_11.txt": [(21, 38, 7), (7, 39, 21), (32, 18, 15), (15, 17, 31), (53, 22, 24), (24, 22, 53),
                         (28, 22, 4), (4, 22, 28), (18, 8, 46), (46, 8, 18), (9, 35, 9)],
    "suncg9_3.txt": [(22, 16, 22), (8, 8, 6), (12, 20, 12)],
    "suncg9_6.txt": [(12, 40, 12), (8, 40, 60), (38, 16, 12), (62, 40, 8), (46, 20, 44), (14, 16, 38)],
}


class _Node(dict):
    __getattr__ = dict.__getitem__

    def __setattr__(self, k, v):
        self[k] = v

    def __deepcopy__(self, memo):
        return _Node({k: copy.deepcopy(v, memo) for k, v in self.items()})


def scannet_benchmark_cfg():
    """experiments/cfgs/ScanNet/benchmark.yml + lib/utils/config.py defaults."""
    c = _Node()
    c.TEST = _Node(RPN_PRE_NMS_TOP_N=400, RPN_POST_NMS_TOP_N=200, RPN_NMS_THRESH=0.1)
    c.NET = "ScanNet_Backbone"
    c.MASK_BACKBONE = "MaskBackbone"
    c.NUM_CLASSES = 19            # main.py:41-50 from nyu40labels_scannet.csv
    c.BATCH_SIZE = 1
    c.RPN_CHANNELS = 256
    c.CLASS_POOLING_SIZE = 4
    c.ALLOW_BORDER = 0
    c.NUM_ANCHORS_LEVEL1 = 3
    c.NUM_ANCHORS_LEVEL2 = 11
    c.NUM_ANCHORS_LEVEL3 = 0
    c.ANCHORS_TYPE_LEVEL1 = "scannet14_3.txt"
    c.ANCHORS_TYPE_LEVEL2 = "scannet14_11.txt"
    c.ANCHORS_TYPE_LEVEL3 = ""
    c.FILTER_ANCHOR_LEVEL1 = ""
    c.FILTER_ANCHOR_LEVEL2 = ""
    c.FILTER_ANCHOR_LEVEL3 = ""
    c.USE_BACKBONE = True
    c.USE_RPN = True
    c.USE_CLASS = True
    c.USE_MASK = True
    c.USE_IMAGES = False          # geometry-only by default here (BASELINE configs 1-3)
    c.ONLY_IMAGES = False
    c.USE_IMAGES_GT = True        # feature maps handed in directly (network.py:199-201)
    c.NUM_2D_CLASSES = 41         # lib/utils/config.py:188
    c.PRETRAINED_ENET_PATH = ""   # benchmark.yml:114 points at scannetv2_enet.pth (not available offline: default init then)
    c.MASK_USE_IMAGES = False
    c.MASK_ONLY_IMAGES = False
    c.NUM_IMAGE_CHANNELS = 128
    c.NUM_IMAGES = 5
    c.DEPTH_SHAPE = [41, 32]
    c.IMAGE_SHAPE = [328, 256]
    c.INTRINSIC = [[37.01983, 0, 20, 0], [0, 38.52470, 15.5, 0], [0, 0, 1, 0], [0, 0, 0, 1]]
    c.PROJ_DEPTH_MIN = 0.1
    c.PROJ_DEPTH_MAX = 4.0
    c.VOXEL_SIZE = 0.046875
    c.CLASS_THRESH = 0.5
    c.MASK_THRESH = 0.5
    c.MAX_VOLUME = 2000000
    c.MAX_IMAGE = 400
    c.TEST_SAVE_DIR = ""
    c.TRUNCATED = 3.0
    # data side (lib/datasets): benchmark.yml:48-51,97-107, config.py:136-141,190
    c.KEEP_THRESH = 1.0
    c.LABEL_MAP = ""              # benchmark.yml points at datagen/fileLists/nyu40labels_scannet.csv (a reference-tree path)
    c.FLIP_TSDF = False
    c.LOG_TSDF = False
    c.MODE = "benchmark"
    c.RANDOM_NUM_IMAGES = False
    c.BASE_IMAGE_PATH = "/mnt/local_datasets/ScanNet/frames_square"
    c.IMAGE_TYPE = "color"
    c.IMAGE_EXT = ".jpg"
    c.COLOR_MEAN = [0.496342, 0.466664, 0.440796]
    c.COLOR_STD = [0.277856, 0.28623, 0.291129]
    return c


def suncg_cfg():
    """experiments/cfgs/SUNCG/rpn_class_mask_5.yml (secondary shape set)."""
    c = scannet_benchmark_cfg()
    c.NET = "SUNCG_Backbone"
    c.NUM_CLASSES = 26
    c.NUM_ANCHORS_LEVEL1 = 3
    c.NUM_ANCHORS_LEVEL2 = 6
    c.ANCHORS_TYPE_LEVEL1 = "suncg9_3.txt"
    c.ANCHORS_TYPE_LEVEL2 = "suncg9_6.txt"
    return c


cfg = scannet_benchmark_cfg()


def anchor_sizes(cfg_, level):
    name = cfg_["ANCHORS_TYPE_LEVEL%d" % level]
    if name in ANCHOR_SETS:
        return ANCHOR_SETS[name]
    # a file path, as the reference's experiments/anchors/<name>
    out = []
    with open(name) as f:
        for line in f:
            if line.strip():
                out.append(tuple(float(v) for v in line.strip().split(",")))
    return out
