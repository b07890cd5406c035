"""A reduced seed sweep on every driver run: tools/parity_sweep.py's cases -- BASELINE configs 1/2 (geometry
chunk, full detection pass), 3 (5-view image path), 4 (4-chunk overlapping scene + whole-scene NMS) -- on FOUR seeds the committed
tests do not use (weights and inputs both change with the seed).  Near-ties are REPORTED, hard mismatches and errors above the
tolerance fail: the same rule as the 32-seed sweep under profiles/ (SURVEY 8c(3)), so the driver's own run carries the evidence
instead of a builder-run file."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

SEEDS = [900, 901, 902, 903]


@pytest.mark.parametrize("case", ["geometry", "images", "scene"])
def test_seed_sweep_has_no_hard_mismatch(case, oracle):
    import parity_sweep as ps
    from parity import report
    fn = {"geometry": lambda s: ps.chunk_case(s, False), "images": lambda s: ps.chunk_case(s, True), "scene": ps.scene_case}[case]
    rows = [fn(s) for s in SEEDS]
    hard = sum(r["hard"] for r in rows)
    near = sum(r["near"] for r in rows)
    worst = max(r["max_err"] for r in rows)
    assert hard == 0, [(s, r) for s, r in zip(SEEDS, rows) if r["hard"]]
    assert worst <= ps.TOL, [(s, r["max_err"]) for s, r in zip(SEEDS, rows)]
    # rows compared down to the classifier agree on the class ids; a scene without near-ties keeps the oracle's keep list
    assert all(r.get("cls_pred_equal", True) for r in rows)
    assert all(r["keep_equal"] for r in rows if "keep_equal" in r and r["near"] == 0)
    report("seed sweep %s: seeds %s, %d proposals, %d matched exactly, %d near-ties, 0 hard, largest error %.3g, %d order swaps"
           % (case, SEEDS, sum(r["oracle"] for r in rows), sum(r["matched"] for r in rows), near, worst, sum(r["swaps"] for r in rows)))
