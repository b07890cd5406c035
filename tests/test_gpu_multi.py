"""The first multi-GPU box tests itself.  When the box has >= 2 (>= 8) GPUs this launches 2 (8)
RCCL ranks of bench.py through `bench.launch_command` -- exactly what the driver runs -- on the 32-chunk overlapping scene
(BASELINE config[4]: chunk c -> rank c mod W, one all_gather_into_tensor of the record blocks over xGMI, whole-scene NMS on every
rank) and requires rank 0's gathered table and keep list to be BIT-IDENTICAL to the one-process result on the same scene (which
tests/test_gpu_scene.py::test_config5_scene_32_chunks_vs_oracle pins to the oracle), and the keep list to be the oracle NMS of the
table.  Skipped with the reason on a 1-GPU lease; SIS3D_BENCH_SHARE_GPU=1 runs the same test with all ranks on GPU 0 and gloo in
place of RCCL (a functional run of the N-rank code path: profiles/r06_multi_rank_shared_gpu.txt)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _one_process(stride, n_chunks):
    from sis3d import config, synthetic
    from sis3d.nets import backbones
    from sis3d.scene import SceneRunner
    import bench
    net, cfg, _ = bench.build_net("scene")
    runner = SceneRunner(net, synthetic.CHUNK_DIMS, solo=True)
    chunks = [(c, bench.scene_origin(c, stride), synthetic.synth_chunk(c).cuda()) for c in range(n_chunks)]
    recs, keep = runner.infer(chunks)
    torch.cuda.synchronize()
    return recs.cpu(), keep.cpu(), cfg


@pytest.mark.parametrize("world", [2, 8])
def test_rccl_ranks_reproduce_the_one_process_scene(world, tmp_path, oracle):
    import bench
    share = bool(os.environ.get("SIS3D_BENCH_SHARE_GPU"))
    have = torch.cuda.device_count()
    if have < world and not share:
        pytest.skip("needs %d GPUs, this box has %d (SIS3D_BENCH_SHARE_GPU=1: every rank on GPU 0 with gloo, a functional run)" % (world, have))
    if share and world > 2:
        pytest.skip("the shared-GPU functional run is made with 2 ranks")
    out = str(tmp_path / ("scene_w%d.npz" % world))
    argv = ["--gpus", str(world), "--workload", "scene", "--steps", "2", "--warmup", "1", "--scene-steps", "2", "--no-cpu-baseline",
            "--no-live-pmc", "--no-side-workloads", "--no-streamed", "--no-calibrate", "--dump-scene", out]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    p = subprocess.run(bench.launch_command(world, argv), env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                       timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == world and line["scaling"] == "strong" and line["config"]["chunks_on_this_rank"] == 32 // world
    got = np.load(out)
    assert int(got["world"]) == world and int(got["n_chunks"]) == 32
    recs, keep, cfg = _one_process(float(got["stride"]), 32)
    # bit-identical to the one-process scene: same per-chunk graphs, the collective only moves the blocks, the merge is deterministic
    assert got["recs"].shape == tuple(recs.shape) and np.array_equal(got["recs"], recs.numpy())
    assert np.array_equal(got["keep"], keep.numpy())
    assert 0 < keep.numel() < recs.shape[0]                      # the overlapping scene: the whole-scene NMS really suppresses
    # and the keep list is the oracle's greedy NMS of the gathered table (integer-exact)
    assert torch.equal(torch.from_numpy(got["keep"]), oracle.nms(torch.from_numpy(got["recs"])[:, :6].contiguous(), cfg.TEST.RPN_NMS_THRESH))
    print("[parity] %d %s ranks: %d records gathered, %d kept -- bit-identical to the one-process scene, keep list = oracle NMS"
          % (world, "gloo (shared GPU)" if share else "RCCL", recs.shape[0], keep.numel()))
