"""Round-5 host logic that needs no GPU: the hardware-queue check of the chunk pipelines, the thread-local dispatch regime, the
lazy scene result, the CPU-baseline thread rule and the per-config sub-objects of the bench line."""
import os
import sys
import threading
import warnings

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_import_leaves_the_environment_alone():
    """r6: `import sis3d` does not edit os.environ; the pipelines' queues are verified by the engine instead"""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
    code = ("import sys, os; sys.path.insert(0, %r); before = dict(os.environ); import sis3d; "
            "print(os.environ.get('GPU_MAX_HW_QUEUES'), sis3d.HW_QUEUES_SET_BY_IMPORT, dict(os.environ) == before)" % os.path.join(ROOT, "3d-sis_amd"))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert out.stdout.split() == ["None", "False", "True"], (out.stdout, out.stderr[-500:])


def test_check_hw_queues_warns_or_raises(monkeypatch):
    from sis3d import engine
    from sis3d._lib import Sis3dError
    monkeypatch.setenv("GPU_MAX_HW_QUEUES", "8")
    assert engine.check_hw_queues(4) is True and engine.default_pipelines() == 4
    monkeypatch.setenv("GPU_MAX_HW_QUEUES", "4")
    assert engine.default_pipelines() == 3
    assert engine.check_hw_queues(3) is True                          # three pipelines run as well on HIP's default of four queues
    assert engine.check_hw_queues(4, verified=True) is True           # r6: four own streams verified on four distinct queues
    engine._QUEUE_WARNED.clear()
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        assert engine.check_hw_queues(4, verified=False) is False
        assert engine.check_hw_queues(4, verified=False) is False     # warned once per count
    assert len(w) == 1 and "export GPU_MAX_HW_QUEUES=8" in str(w[0].message)
    with pytest.raises(Sis3dError) as ei:
        engine.check_hw_queues(4, strict=True, verified=False)
    assert "BEFORE it starts" in str(ei.value)
    monkeypatch.setenv("SIS3D_STRICT_HW_QUEUES", "1")
    with pytest.raises(Sis3dError):
        engine.check_hw_queues(5)


def test_dispatch_regime_is_thread_local():
    from sis3d import ops
    seen = {}
    gate = threading.Barrier(2)

    def worker(name, shared, cap):
        with ops.dispatch_regime(shared, cap):
            gate.wait()                                               # both threads are inside their blocks at the same time
            seen[name] = ops.regime()
            gate.wait()
        seen[name + "_after"] = ops.regime()
    ts = [threading.Thread(target=worker, args=("a", True, 108)), threading.Thread(target=worker, args=("b", False, 0))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert seen == {"a": (True, 108), "b": (False, 0), "a_after": (False, 0), "b_after": (False, 0)}
    assert ops.regime() == (False, 0)
    # the tally is per thread too
    ops.flop_tally(True)
    other = []
    t = threading.Thread(target=lambda: other.append(ops.flop_tally(False)))
    t.start(); t.join()
    assert other == [None] and ops.flop_tally(False) == {"wino_algorithmic_flops": 0.0, "wino_launches": 0}


def test_no_process_wide_dispatch_setters_left():
    """the regime is an argument, not a global -- neither the header nor the library exports a setter"""
    hdr = open(os.path.join(ROOT, "include", "sis3d.h")).read()
    for name in ("sis3d_conv3d_k3wino_set_shared_chip", "sis3d_conv3d_k3t16_set_brick_cap", "sis3d_nms_set_path"):
        assert name + "(" not in hdr
    for f in ("conv3d_wino.hip", "conv3d_t16.hip", "nms.hip"):
        src = open(os.path.join(ROOT, "3d-sis_amd", "csrc", f)).read()
        assert "g_shared_chip" not in src and "g_brick_cap" not in src and "g_nms_path" not in src
    assert "SIS3D_DISPATCH_SHARED_CHIP" in hdr


def test_scene_result_resolved_wraps_eager_tuples():
    from sis3d.scene import SceneResult
    recs, keep = torch.zeros(3, 16), torch.tensor([0, 2])
    r = SceneResult.resolved((recs, keep))
    assert r.resolve()[0] is recs and r.resolve()[1] is keep
    r3 = SceneResult.resolved((recs, keep, {0: "m"}))
    assert len(r3.resolve()) == 3


def test_cpu_thread_rule_and_config_entry(monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    n, why = bench.cpu_threads_rule()
    assert 1 <= n <= 64 and isinstance(why, str)
    monkeypatch.setenv("SIS3D_CPU_THREADS", "7")
    assert bench.cpu_threads_rule() == (7, "SIS3D_CPU_THREADS")
    res = {"dt": 0.04, "vox_per_step": 4 * bench.VOXELS, "single_ms": 0.4, "wino_flops": 30e9,
           "extra": {"mask_head_gflop": 37.75, "mask_boxes": 16, "enet_ms_5_views": 0.2}}
    e = bench.config_entry(res, 20, "detect", masks=True)
    assert abs(e["value"] - 4 * bench.VOXELS * 20 / 0.04) < 1 and e["ms_per_step"] == 2.0
    assert abs(e["algorithmic_gflop_per_chunk"] - (43.48 + 37.75)) < 1e-6 and 0 < e["fp32_frac"] < 1 and 0 < e["hbm_frac"] < 1
    assert e["mask_boxes"] == 16 and e["single_chunk_latency_ms"] == 0.4
    assert bench.fracs_above_one(e) == []
    st = bench.streamed_entry({"dt": 0.025, "bytes_per_chunk": 2 * bench.VOXELS * 4, "ring": 4}, 0.02, 20, 4 * bench.VOXELS, 4, 1, "grid")
    assert abs(st["ratio_to_resident"] - 0.8) < 1e-9 and abs(st["h2d_gbs_per_gpu"] - 3200 * 3.538944e6 / 1e9) < 1e-6
    t, n_runs, lo, hi = bench._median_runs(lambda: None, 0.0, min_runs=10)
    assert n_runs == 10 and lo <= t <= hi


def test_rank_cpu_blocks_partition_the_host():
    """r5: N launcher processes on one host each get their own block of cores + SMT siblings"""
    from sis3d import parallel
    avail = list(range(256))                                       # 2 sockets x 64 cores x 2 threads
    blocks = [parallel.rank_cpu_block(avail, r, 8) for r in range(8)]
    assert all(len(b) == 32 for b in blocks) and sorted(c for b in blocks for c in b) == avail
    assert blocks[0][:16] == list(range(16)) and blocks[0][16:] == list(range(128, 144))     # cores 0-15 and their siblings
    assert blocks[4][:16] == list(range(64, 80))                                             # ranks 4-7 on the second socket
    assert parallel.rank_cpu_block(avail, 0, 1) == avail
    assert parallel.rank_cpu_block(list(range(8)), 1, 8) == list(range(8))                   # too few CPUs to split: left alone
    odd = parallel.rank_cpu_block(list(range(0, 48)), 2, 4)
    assert len(odd) == 12 and len(set(odd)) == 12
