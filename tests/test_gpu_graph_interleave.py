"""replays of the captured IMAGE-path graphs interleaved with eager launches of the library and host synchronisations.
Rounds 1-2 saw "Memory access fault by GPU" in exactly this loop; the root cause (tools/hipgraph_memset_repro.py, no sis3d code) is that
a hipMemsetAsync captured into a HIP graph replays, from the second replay on, with a garbage fill pattern on ROCm 7.2 -- the
voxel->pixel table of the back-projection was cleared that way, so later replays gathered through garbage indices.  Every clear is now
a kernel of the library (csrc/api.hip sis3d_fill32); this test is the loop that used to fault, 1000 replays per graph, and the outputs
of every checked replay must equal, bit for bit, the same pass launched EAGERLY (never replayed) on the same inputs."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from sis3d import config, ops, synthetic  # noqa: E402


def _net(use_images):
    from sis3d.nets import backbones
    cfg = config.scannet_benchmark_cfg()
    cfg.USE_IMAGES = use_images
    cfg.USE_MASK = False
    net = backbones.ScanNet_Backbone(cfg=cfg)
    net.init_modules()
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    net.load_state_dict(synthetic.synth_state_dict(shapes, seed=0, gains=synthetic.DEFAULT_GAINS))
    return net.cuda().eval()


def _snap(eng):
    return [{k: v.detach().clone() for k, v in e.out.items() if torch.is_tensor(v)} for e in eng.engines]


def test_captured_memset_free_library():
    """no source of the library may call hipMemsetAsync / hipMemset: a captured one becomes the memset node that replays wrongly"""
    import glob
    import os
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "3d-sis_amd", "csrc")
    srcs = glob.glob(os.path.join(root, "*.hip")) + glob.glob(os.path.join(root, "*.h"))
    assert srcs
    for path in srcs:
        code = "\n".join(ln.split("//")[0] for ln in open(path).read().splitlines())
        assert "hipMemset" not in code, path


@pytest.mark.parametrize("iters", [1000])
def test_image_graph_replays_interleaved_with_eager_launches(iters):
    from sis3d.engine import PipelinedEngines
    net = _net(True)
    dims = (48, 24, 48)
    eng = PipelinedEngines(net, 2, stage="rpn", use_graph=True, dims=dims, n_views=3)
    eager = PipelinedEngines(net, 2, stage="rpn", use_graph=False, dims=dims, n_views=3)
    for i in range(2):
        data = synthetic.synth_chunk(i, dims)
        feats, i3d, i2d = synthetic.synth_views(i, n_views=3, n_per_view=1500, dims=dims)
        eng.load(i, data, feats, i3d, i2d)
        eager.load(i, data, feats, i3d, i2d)
    eager.prepare(warmup=1)                           # eager launches only, under the same brick cap as the captures below
    torch.cuda.synchronize()
    want = _snap(eager)
    assert all(float(w["rpn_cls_prob_level1"].sum()) > 0 for w in want)
    eng.prepare(warmup=2)
    dev = torch.device("cuda")
    x = ops.new_act(128, (24, 12, 24), dev).normal_().clamp_(min=0)
    x32 = ops.new_act(32, (48, 24, 48), dev).normal_().clamp_(min=0)
    pc32 = ops.PackedConv(torch.randn(32, 32, 3, 3, 3, device=dev) * 0.05, torch.zeros(32, device=dev))
    boxes = torch.rand(400, 6, device=dev) * 40
    boxes[:, 3:] += boxes[:, :3] + 1
    checked = 0
    for it in range(iters):
        eng.run()                                     # one replay of each of the two captured graphs, on their own streams
        net.rpn_net_level1(x)                         # Winograd kernel, 148 KB of LDS
        ops.conv3d(x32, pc32, relu=True)              # direct k3t16 kernel
        ops.nms(boxes, 0.3)
        (x32 * 1.5 + 0.25).clamp_(min=0)              # a plain PyTorch kernel
        torch.cuda.synchronize()                      # host synchronise between replays: the round-2 trigger
        if it < 4 or it % 100 == 99:
            for a, b in zip(want, _snap(eng)):
                assert a.keys() == b.keys()
                for k in a:
                    assert torch.equal(a[k], b[k]), (it, k)
            checked += 1
    assert checked >= 10
