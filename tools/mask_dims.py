"""crop windows of the bench's fixed mask detection set (detect --masks): dims, blocks of 8x4x8 and of 4x4x4"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "3d-sis_amd")); sys.path.insert(0, ROOT)
import bench
from sis3d import synthetic
from sis3d.engine import ChunkEngine
net, cfg, sd = bench.build_net("detect", masks=True)
eng = ChunkEngine(net, stage="detect", mask_boxes=16, use_graph=False)
eng.load(synthetic.synth_chunk(0))
eng.prepare(warmup=1)
p = eng.mask_plan
cd = lambda a, b: -(-a // b)
b8 = sum(cd(x, 8) * cd(y, 4) * cd(z, 8) for x, y, z in p.dims)
m4 = sum(cd(x, 4) * cd(y, 4) * cd(z, 4) for x, y, z in p.dims)
print("dims", [tuple(d) for d in p.dims])
print("voxels", p.voxels, "blocks 8x4x8:", b8, "(slots %d)" % (b8 * 256), "minis 4x4x4:", m4, "(slots %d)" % (m4 * 64), "-> WG-equivalents %.1f" % (m4 / 4))
