"""bench.py's N > 1 entry: `python bench.py --gpus N` must itself start N ranks, and must never
report a smaller world than it was asked for.  The launch / rendezvous / record all-gather / max-over-ranks logic runs
here on CPU under gloo (`--selftest-cpu`, the GPU workloads replaced by synthetic record blocks)."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, BENCH] + args, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                       timeout=timeout)
    return p.returncode, p.stdout, p.stderr


SIDE_KEYS = {"chunk_pipeline": ("workload", "value", "unit", "scaling", "ms_per_step", "chunks_per_step_per_gpu"),
             "scene": ("workload", "value", "unit", "scaling", "ms_per_scene", "steps", "scene_chunks", "scene_stride", "records_gathered",
                       "kept_after_scene_nms")}


def _check_line(out, n, steps):
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out                                  # exactly one JSON line, from rank 0
    d = json.loads(lines[0])
    assert d["n_gpus"] == n and d["steps"] == steps and d["selftest"] is True
    assert d["config"]["records"] == sum(3 + c % 5 for c in range(8))     # every chunk of every rank arrived
    assert out.strip().splitlines()[-1] == lines[0]              # ... and it is the last thing on stdout
    # BOTH workloads on every line, at every N, under the same keys, so value(N) / value(1) compares like with like
    for key, fields in SIDE_KEYS.items():
        assert key in d, key
        for f in fields:
            assert f in d[key], (key, f)
    assert d["chunk_pipeline"]["scaling"] == "weak" and d["scene"]["scaling"] == "strong"
    assert d["scene"]["records_gathered"] == d["config"]["records"]
    return d


@pytest.mark.parametrize("n", [1, 2])
def test_default_line_carries_both_workloads_at_every_n(n):
    rc, out, err = _run(["--gpus", str(n), "--steps", "3", "--warmup", "0", "--scene-chunks", "8", "--selftest-cpu"])
    assert rc == 0, err[-2000:]
    d = _check_line(out, n, 3)
    assert d["value"] == d["chunk_pipeline"]["value"] and d["scaling"] == "weak"     # auto: the headline is config[1] at every N


def test_gpus_2_scene_headline_gloo():
    rc, out, err = _run(["--gpus", "2", "--steps", "3", "--warmup", "0", "--scene-chunks", "8", "--workload", "scene", "--selftest-cpu"])
    assert rc == 0, err[-2000:]
    d = _check_line(out, 2, 3)
    assert d["value"] == d["scene"]["value"] and d["scaling"] == "strong"


@pytest.mark.skipif(torch.cuda.is_available() and torch.cuda.device_count() >= 2, reason="box has >= 2 GPUs")
def test_gpus_2_without_two_gpus_fails_loudly():
    rc, out, err = _run(["--gpus", "2", "--steps", "2", "--warmup", "1"])
    assert rc != 0
    assert "refusing" in err and "--gpus 2" in err
    assert not [ln for ln in out.splitlines() if ln.startswith("{")]       # no line claiming n_gpus: 1


def test_world_size_mismatch_is_refused():
    rc, out, err = _run(["--gpus", "2", "--selftest-cpu"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert rc != 0 and "mismatched" in err and "{" not in out


def test_launch_command_is_one_rank_per_gpu_on_loopback():
    sys.path.insert(0, ROOT)
    import bench
    cmd = bench.launch_command(8, ["--gpus", "8", "--steps", "5"], port=29999)
    assert cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29999"
    assert cmd[-4:] == ["--gpus", "8", "--steps", "5"] and cmd[-5].endswith("bench.py")
    # defaults: N = 1, `auto` = config[1] as the headline at every N with the 32-chunk overlapping scene as a side key
    a = bench.parse([])
    assert a.gpus == 1 and a.workload == "auto" and a.scene_chunks == 32 and a.scene_stride == 80.0


def test_roofline_fractions_are_fractions():
    """VERDICT r3 / ADVICE r3: every `*frac*` key of the bench line is executed work over a roof (<= 1); the direct-convolution
    pricing of a Winograd launch lives in flat `algorithmic_*` siblings the driver's `parsed` record keeps."""
    sys.path.insert(0, ROOT)
    import bench
    e = bench.roofline_entry(50.5e-6, 95.5e-6, True)              # round-3 timings of the dominant kernel
    assert abs(e["frac"] - 0.456) < 2e-3 and abs(e["achieved"] - e["frac"] * e["peak"]) < 1e-9
    assert abs(e["flops_per_launch"] * 3.375 - 2.0 * 6912 * 256 * 128 * 27) < 1.0
    assert abs(e["algorithmic_tflops"] - 242.19) < 0.01 and e["algorithmic_speedup_vs_direct_count"] == 3.375
    assert e["traffic_source"].startswith("profiles/") and abs(e["traffic_ratio"] - e["traffic"] / e["algorithmic_bytes_per_launch"]) < 1e-12
    assert not isinstance(e.get("executed"), dict)                 # nothing the headline needs is nested any more
    assert bench.fracs_above_one(e) == []
    d = bench.roofline_entry(95.5e-6, 0.0, False)
    assert abs(d["frac"] - 0.814) < 2e-3 and d["algorithmic_speedup_vs_direct_count"] == 1.0 and bench.fracs_above_one(d) == []
    assert bench.fracs_above_one({"stages": {"rpn": {"fp32_frac": 1.5}}, "x": [{"hbm_frac": 0.2}]}) == [("stages.rpn.fp32_frac", 1.5)]
    assert abs(bench.executed_flops(42.58e9, 30.57e9) - (42.58e9 - 30.57e9 * (1 - 64 / 216))) < 1.0


def test_hardware_queues_and_chunks_in_flight(monkeypatch):
    """r4: bench.py asks HIP for 8 hardware queues before torch is imported (a fourth pipeline otherwise shares a queue with another
    stream) and keeps FOUR chunks in flight on them -- three when the caller pinned HIP's default of 4 (profiles/r04_hw_queues.txt)"""
    sys.path.insert(0, ROOT)
    import bench
    assert os.environ.get("GPU_MAX_HW_QUEUES")                      # set (or already present) as soon as bench is imported
    monkeypatch.setenv("GPU_MAX_HW_QUEUES", "8")
    assert bench.hw_queues() == 8 and all(bench.default_inflight(w) == 4 for w in ("backbone_rpn", "detect", "images", "scene"))
    monkeypatch.setenv("GPU_MAX_HW_QUEUES", "4")
    assert bench.hw_queues() == 4 and bench.default_inflight("backbone_rpn") == 3
    monkeypatch.setenv("GPU_MAX_HW_QUEUES", "many")
    assert bench.hw_queues() == 4
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert src.index('os.environ.setdefault("GPU_MAX_HW_QUEUES"') < src.index("def parse(")        # before anything can import torch
    assert "\nimport torch" not in src.split("def parse(")[0]


def test_cpu_baseline_of_the_rgb_workload_times_the_encoder_too():
    """`bench.py --workload images --rgb`: the CPU baseline's views are RGB images; its stage table starts from the oracle's ENet feature
    maps (r6: it fed the RGB views to the 128-channel colour branch and took the line down)"""
    from sis3d import config, synthetic
    from sis3d.nets import backbones
    from benchlib.cpu_baseline import cpu_baseline
    cfg = config.scannet_benchmark_cfg()
    cfg.USE_IMAGES, cfg.USE_IMAGES_GT, cfg.USE_MASK = True, False, False
    net = backbones.ScanNet_Backbone(cfg=cfg)
    net.init_modules()
    sd = synthetic.synth_checkpoint({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed=0)
    r = cpu_baseline("images", sd, cfg, 1.0)
    assert r["value"] > 0 and r["port"]["value"] > 0
    assert {"enet_encoder_5_views", "rpn_convs_heads", "projection_view_max"} <= set(r["stages"])
