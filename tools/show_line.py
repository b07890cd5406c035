"""print the interesting numbers of a bench.py JSON line (file or stdin)"""
import json
import sys

src = open(sys.argv[1]) if len(sys.argv) > 1 else sys.stdin
d = json.loads([ln for ln in src if ln.startswith("{")][-1])
cp, sc = d.get("chunk_pipeline", {}), d.get("scene", {})
print("value %.3f G voxels/s  ms/step %.3f  single %.3f ms  roofline frac %.3f (%.1f us)  step fp32 %.3f hbm %.3f" % (
    d["value"] / 1e9, d["ms_per_step"], d["config"].get("single_chunk_latency_ms") or 0, d["roofline"]["frac"], d["roofline"]["launch_us"],
    d["step_roofline"].get("fp32_frac", 0), d["step_roofline"]["hbm_frac"]))
print("  stream window", d["config"].get("stream_window"), d["config"].get("stream_window_ms_per_step"))
for k in ("streamed", "streamed_sdf"):
    if k in cp:
        e = cp[k]
        print("  chunk_pipeline.%s: %s" % (k, ("%.3f G  %.3f ms/step  ratio %.3f  h2d %.1f GB/s" % (e["value"] / 1e9, e["ms_per_step"], e["ratio_to_resident"], e["h2d_gbs_per_gpu"])) if "value" in e else e))
if sc:
    print("scene %.3f ms  (%.3f G)  kept %s/%s  one-launch %s" % (sc["ms_per_scene"], sc["value"] / 1e9, sc.get("kept_after_scene_nms"), sc.get("records_gathered"), sc.get("one_graph_launch_per_scene")))
    if "calibration" in sc:
        print("  calibration", sc["calibration"])
    if "streamed" in sc:
        e = sc["streamed"]
        print("  scene.streamed: %s" % (("%.3f ms  ratio %.3f" % (e["ms_per_scene"], e["ratio_to_resident"])) if "value" in e else e))
    sh = sc.get("share_of_one_rank_at_8")
    if sh:
        print("  share: %.3f ms + collective %s us -> ceiling %.2f (without collective %.2f); one-launch %s" % (
            sh["ms"], sh.get("collective_us_world_of_one"), sh["ceiling_speedup_at_8"], sh.get("ceiling_without_collective_term", 0), sh.get("one_graph_launch_per_scene")))
for k in ("detect", "detect_masks", "images", "images_rgb"):
    if k in d:
        e = d[k]
        print("%-13s %s" % (k, ("%.3f G  %.3f ms/step  single %.3f ms  fp32 %.3f hbm %.3f %s" % (
            e["value"] / 1e9, e["ms_per_step"], e["single_chunk_latency_ms"] or 0, e.get("fp32_frac", 0), e["hbm_frac"],
            ("enet %.3f ms" % e["enet_ms_5_views"]) if "enet_ms_5_views" in e else ("mask head %.3f ms" % e["mask_head_ms"]) if "mask_head_ms" in e else "")) if "value" in e else e))
if "stages" in d:
    st = d["stages"]
    print("stages: backbone %.3f ms (fp32 %.3f hbm %.3f)  rpn %.3f ms (fp32 %.3f)" % (st["backbone"]["ms"], st["backbone"]["fp32_frac"], st["backbone"]["hbm_frac"], st["rpn"]["ms"], st["rpn"]["fp32_frac"]))
cb = d.get("cpu_baseline")
if cb:
    print("cpu: %.2f M voxels/s kind %s cores %s  (port %.2f M)" % (cb["value"] / 1e6, cb["kind"], cb["cores"], cb.get("port", {}).get("value", 0) / 1e6))
