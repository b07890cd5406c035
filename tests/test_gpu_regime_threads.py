"""r5 ("thread-safe w.r.t. distinct streams"): the dispatch regime of the k3 convolutions is a
per-call argument of the C ABI (SIS3D_DISPATCH_SHARED_CHIP / the *_prefer functions' shared_chip / sis3d_conv3d_k3t16_brick's
max_voxels) and a THREAD-LOCAL value on the Python side (ops.dispatch_regime).  Two threads running the two regimes at the same
time, each on its own stream, must launch exactly what each regime launches when run alone: same number of Winograd launches,
bit-identical outputs."""
import threading

import pytest
import torch

pytestmark = pytest.mark.gpu

from sis3d import config, synthetic  # noqa: E402


def _net():
    from sis3d.nets import backbones
    cfg = config.scannet_benchmark_cfg()
    net = backbones.ScanNet_Backbone(cfg=cfg)
    net.init_modules()
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    net.load_state_dict(synthetic.synth_state_dict(shapes, seed=0, gains=synthetic.DEFAULT_GAINS))
    return net.cuda().eval()


def _pass(net, scene, shared, cap):
    from sis3d import ops
    with ops.dispatch_regime(shared_chip=shared, brick_cap=cap), torch.no_grad():
        ops.flop_tally(True)
        try:
            net.backbone_rpn(scene)
        finally:
            t = ops.flop_tally(False)
    out = {k: v.clone() for k, v in net._predictions.items() if k.startswith("rpn_") and torch.is_tensor(v)}
    return t["wino_launches"], out


def test_two_threads_two_regimes_launch_what_each_launches_alone():
    from sis3d import ops
    assert ops.regime() == (False, 0)
    regimes = [(False, 0), (True, 108)]
    nets = [_net(), _net()]                                 # one replica per thread (Network keeps per-forward state in attributes)
    scene = synthetic.synth_chunk(3).cuda()
    serial = []
    for net, (sh, cap) in zip(nets, regimes):
        _pass(net, scene, sh, cap)                          # warm-up: weight packs
        serial.append(_pass(net, scene, sh, cap))
    torch.cuda.synchronize()
    assert serial[1][0] > serial[0][0], "the shared-chip regime sends more layers to the Winograd kernel"
    assert ops.regime() == (False, 0)                       # nothing process-wide was left behind

    start = threading.Barrier(2)
    results, errors = [None, None], []

    def worker(i):
        try:
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.default_stream())
            with torch.cuda.stream(s):
                start.wait()
                got = []
                for _ in range(12):                         # many interleaved passes: a shared setting would be seen by the other thread
                    got.append(_pass(nets[i], scene, *regimes[i]))
                s.synchronize()
            results[i] = got
        except Exception as e:                              # pragma: no cover
            errors.append((i, repr(e)))
            try:
                start.abort()
            except Exception:
                pass

    ts = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors
    for i in range(2):
        for launches, out in results[i]:
            assert launches == serial[i][0], (i, launches, serial[i][0])
            for k, v in serial[i][1].items():
                assert torch.equal(out[k], v), (i, k)
    print("[parity] regimes in two threads: serial Winograd launches %d / %d, 12 concurrent passes each bit-identical"
          % (serial[0][0], serial[1][0]))


def test_regime_nests_and_restores():
    from sis3d import ops
    assert ops.regime() == (False, 0)
    with ops.dispatch_regime(True, 108):
        assert ops.regime() == (True, 108)
        with ops.dispatch_regime(False, 0):
            assert ops.regime() == (False, 0)
        assert ops.regime() == (True, 108)
    assert ops.regime() == (False, 0)
