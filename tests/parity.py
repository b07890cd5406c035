"""End-to-end proposal parity, the protocol of SURVEY.md 8(c)(3): the device and oracle proposal lists are compared as
SETS with exact (IoU = 1, coordinates within `tol`) one-to-one matching; a box left unmatched on either side FAILS unless
its RPN score sits within `gap` of another candidate's score (a near-tie: the two fp32 pipelines differ by ~1e-6 in the
logits, so the stable sort may order such a pair differently and NMS then keeps the other one) -- those are REPORTED.

Contract of the committed seeds (round 3): every end-to-end test passes `max_near=0` (the default), i.e. a near-tie FAILS the
test; the row-for-row comparisons that follow are therefore unconditional.  Should a box ever produce a near-tie on some host,
the remedy is a different seed for that case, stated in the test -- not a silent skip.  Every `[parity]` line is also appended
to the file named by $SIS3D_PARITY_LOG (profiles/r03_parity_log.txt is one such run on the MI355X box).

How often near-ties occur on seeds that were NOT chosen is measured, not assumed: tools/parity_sweep.py runs 32 fresh seeds (weights
and inputs) through configs 1/2, 3 and 4 with `compare_proposals` below -- the same comparison without the assertions -- and writes
the counts of exact matches, near-ties and hard mismatches to profiles/r05_parity_sweep.txt (round 5)."""
import os

import torch


def report(line):
    """print a `[parity]` line and append it to $SIS3D_PARITY_LOG when set (pytest -q swallows stdout)"""
    line = "[parity] " + line
    print(line)
    path = os.environ.get("SIS3D_PARITY_LOG")
    if path:
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        with open(path, "a") as f:
            f.write(line + "\n")


def match_sets(got, want, tol=1e-3):
    """greedy one-to-one matching of box rows by max-abs coordinate difference <= tol -> (pairs [(i_want, j_got)],
    unmatched_want, unmatched_got)"""
    pairs, used = [], set()
    if len(want) and len(got):
        d = (got[None, :, :] - want[:, None, :]).abs().amax(-1)           # (W, G)
        for i in range(d.shape[0]):
            row = d[i].clone()
            if used:
                row[list(used)] = float("inf")
            j = int(row.argmin())
            if float(row[j]) <= tol:
                pairs.append((i, j))
                used.add(j)
    mw = {i for i, _ in pairs}
    return pairs, [i for i in range(len(want)) if i not in mw], [j for j in range(len(got)) if j not in used]


def nearest_score_gap(score, all_scores_sorted):
    """distance from `score` to the closest OTHER candidate score (all_scores_sorted: every candidate, descending)"""
    a = all_scores_sorted
    d = (a - score).abs()
    k = int(d.argmin())                                     # the candidate itself (or its fp32 twin on the other side)
    d[k] = float("inf")
    return float(d.min()) if d.numel() > 1 else float("inf")


def compare_proposals(got_rois, got_scores, want_rois, want_scores, all_scores_sorted, tol=1e-3, gap=1e-5):
    """the comparison of assert_proposals_match WITHOUT the assertions (tools/parity_sweep.py: near-tie frequency over many seeds,
    reported instead of selected away -- SURVEY.md 8c(3)) -> dict(oracle, device, matched, near, hard, max_score_err): `near` =
    unmatched boxes whose RPN score has another candidate's score within `gap`; `hard` = unmatched boxes WITHOUT such a rival, i.e.
    real disagreements"""
    got_rois, want_rois = got_rois.float().cpu(), want_rois.float().cpu()
    gs, ws = got_scores.float().cpu().view(-1), want_scores.float().cpu().view(-1)
    pairs, un_w, un_g = match_sets(got_rois, want_rois, tol)
    err = max([abs(float(ws[i]) - float(gs[j])) for i, j in pairs] or [0.0])
    near = hard = 0
    for idx, sc in ((un_w, ws), (un_g, gs)):
        for k in idx:
            if nearest_score_gap(float(sc[k]), all_scores_sorted.float().cpu().clone()) <= gap + 2e-6:
                near += 1
            else:
                hard += 1
    # same SET, different ORDER: the lists are score-ordered, so two kept proposals whose scores tie within fp32 noise may swap places
    # without anything being kept or dropped differently.  swaps = matched pairs sitting at different positions; swap_gap = the
    # largest oracle-score difference between the two positions such a pair occupies (a benign swap has a gap of ~1e-7)
    swaps = [(i, j) for i, j in pairs if i != j]
    swap_gap = max([abs(float(ws[i]) - float(ws[j])) for i, j in swaps if j < len(ws)] or [0.0])
    return {"oracle": len(want_rois), "device": len(got_rois), "matched": len(pairs), "near": near, "hard": hard, "max_score_err": err,
            "swaps": len(swaps), "swap_gap": swap_gap, "pairs": pairs}


def assert_proposals_match(got_rois, got_scores, want_rois, want_scores, all_scores_sorted, tol=1e-3, gap=1e-5, score_tol=1e-4,
                           label="", max_near=0):
    """-> number of near-ties (asserted <= max_near; 0 for every committed seed).  got_* device outputs (CPU tensors),
    want_* the oracle's, all_scores_sorted the oracle's full candidate score list (OracleNet.forward()['_scores_sorted_all'])."""
    got_rois, want_rois = got_rois.float().cpu(), want_rois.float().cpu()
    gs, ws = got_scores.float().cpu().view(-1), want_scores.float().cpu().view(-1)
    pairs, un_w, un_g = match_sets(got_rois, want_rois, tol)
    for i, j in pairs:
        assert abs(float(ws[i]) - float(gs[j])) <= score_tol, (label, "score of a matched box", i, j, float(ws[i]), float(gs[j]))
    near = 0
    for side, idx, sc in (("oracle-only", un_w, ws), ("device-only", un_g, gs)):
        for k in idx:
            g_ = nearest_score_gap(float(sc[k]), all_scores_sorted.float().cpu().clone())
            assert g_ <= gap + 2e-6, "%s %s box %d (score %.7f) has no near-tied rival: nearest other candidate score is %.3g away" % (
                label, side, k, float(sc[k]), g_)
            near += 1
    report("%s: %d oracle / %d device proposals, %d matched exactly, %d near-ties" % (
        label, len(want_rois), len(got_rois), len(pairs), near))
    assert near <= max_near, "%s: %d near-tied proposals (allowed %d): pick a seed without near-ties for this case" % (label, near, max_near)
    return near
