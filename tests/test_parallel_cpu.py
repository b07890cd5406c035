"""N>1 path on CPU: world_size-2 (and 3) gloo processes run the chunk sharding / record all-gather /
whole-scene NMS logic of sis3d.parallel with the CPU ORACLE injected as detector and NMS (test-only);
the result must equal the single-process run and be identical on every rank."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_detect(payload):
    """deterministic pseudo-detections per chunk (stands in for the per-chunk pipeline)"""
    from sis3d.engine import RECORD_WIDTH
    cid = payload
    g = torch.Generator().manual_seed(100 + cid)
    k = 12
    n = 5 + cid % 7
    lo = torch.rand(k, 3, generator=g) * torch.tensor([80.0, 40.0, 80.0])
    rec = torch.zeros(k, RECORD_WIDTH)
    rec[:, :3] = lo
    rec[:, 3:6] = lo + torch.rand(k, 3, generator=g) * 30 + 1
    rec[:, 6] = torch.rand(k, generator=g).round(decimals=1)     # coarse scores -> many ties across chunks
    rec[:, 7] = 1 + (cid % 2)
    rec[:, 8] = torch.randint(1, 19, (k,), generator=g).float()                  # class id
    rec[:, 9] = torch.rand(k, generator=g)                                       # class probability, both sides of 0.5
    rec[:, 10:16] = rec[:, :6] + (torch.rand(k, 6, generator=g) - 0.5) * 4       # class-regressed box
    rec[0, 13] = rec[0, 10] + 0.3                                                # collapses when rounded to voxels
    return rec, n


def _fake_masks(payload, windows, classes):
    """stands in for the mask head: a tensor determined by (chunk, window, class)"""
    return [torch.full((w[3] - w[0], w[4] - w[1], w[5] - w[2]), float(100 * payload + k)) for w, k in zip(windows, classes)]


def _chunks(n):
    return [(c, (96.0 * (c % 4), 0.0, 96.0 * (c // 4)), c) for c in range(n)]


def _worker(rank, world, port, n_chunks, out_dir):
    for p in (os.path.join(ROOT, "3d-sis_amd"), os.path.join(ROOT, "oracle")):
        sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sis3d import parallel
    import sis3d_oracle as orc
    recs, keep, masks = parallel.infer_scene(_chunks(n_chunks), _fake_detect, orc.nms, 12, 0.1, mask_fn=_fake_masks)
    torch.save((recs, keep, masks), os.path.join(out_dir, "r%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_chunks", [(2, 8), (2, 5), (3, 7)])
def test_sharded_scene_equals_single_process(tmp_path, world, n_chunks, oracle):
    from sis3d import parallel
    want_recs, want_keep, want_masks = parallel.infer_scene(_chunks(n_chunks), _fake_detect, oracle.nms, 12, 0.1,
                                                            mask_fn=_fake_masks)
    assert want_recs.shape[0] == sum(5 + c % 7 for c in range(n_chunks))
    # single process: exactly the confident, non-degenerate survivors carry a mask, cropped in scene coordinates
    kept = want_recs[want_keep]
    assert 0 < len(want_masks) < kept.shape[0] and set(want_masks) <= set(range(kept.shape[0]))
    for i, (w, m) in want_masks.items():
        assert tuple(m.shape) == (w[3] - w[0], w[4] - w[1], w[5] - w[2]) and kept[i, 9] > 0.5
        assert float(m.flatten()[0]) % 100 == kept[i, 8]
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_chunks, str(tmp_path)), nprocs=world, join=True)
    union = {}
    for r in range(world):
        recs, keep, masks = torch.load(os.path.join(str(tmp_path), "r%d.pt" % r))
        assert torch.equal(recs, want_recs), r
        assert torch.equal(keep, want_keep), r
        assert not set(masks) & set(union)                         # every detection is masked by exactly one rank
        for i, (w, m) in masks.items():
            assert int(float(m.flatten()[0])) // 100 % world == r   # ... the owner of the chunk that produced it
        union.update(masks)
    assert set(union) == set(want_masks)
    for i in union:
        assert union[i][0] == want_masks[i][0] and torch.equal(union[i][1], want_masks[i][1])


def test_pack_and_merge_rules(oracle):
    from sis3d import parallel
    from sis3d.engine import RECORD_WIDTH
    assert parallel.shard_chunks(7, 1, 3) == [1, 4]
    rec, n = _fake_detect(3)
    blk = parallel.pack_block(rec, torch.tensor([n], dtype=torch.int32), (96.0, 0.0, 192.0))
    assert blk.numel() == parallel.block_floats(12) and blk[0] == n
    rows = blk[1:].view(12, RECORD_WIDTH)
    assert torch.equal(rows[:n, 0], rec[:n, 0] + 96.0) and torch.equal(rows[:n, 5], rec[:n, 5] + 192.0)
    assert (rows[n:] == 0).all()
    # ties: stable order = (chunk,row)
    blocks = torch.stack([parallel.pack_block(*_fake_detect(c), (0.0, 0.0, 0.0)) for c in range(3)])
    recs, keep = parallel.merge_scene(blocks, 12, oracle.nms, 0.1)
    s = recs[:, 6]
    assert (s[:-1] >= s[1:]).all()
    assert torch.equal(keep, oracle.nms(recs[:, :6].contiguous(), 0.1))


def test_merge_fn_contract():
    """parallel.merge_scene(merge_fn=...): a fused implementation may take the table or refuse it (None -> the torch code runs);
    either way the caller sees (records, keep[, chunk ids])"""
    import sis3d_oracle as orc
    from sis3d import parallel
    from sis3d.engine import RECORD_WIDTH as W
    g = torch.Generator().manual_seed(3)
    k_rows, n_chunks = 5, 3
    blocks = torch.zeros(n_chunks, 1 + k_rows * W)
    for c, n in enumerate((5, 0, 3)):
        rows = torch.zeros(k_rows, W)
        lo = torch.rand(n, 3, generator=g) * 50 + 60 * c
        rows[:n, 0:3], rows[:n, 3:6] = lo, lo + 10
        rows[:n, 6] = torch.rand(n, generator=g)
        blocks[c, 0], blocks[c, 1:] = n, rows.reshape(-1)
    want = parallel.merge_scene(blocks, k_rows, orc.nms, 0.1, with_chunk_ids=True)
    calls = []

    def refuse(b, k, t, sc, bc, mk):
        calls.append((k, t, sc, bc, mk))
        return None

    got = parallel.merge_scene(blocks, k_rows, orc.nms, 0.1, with_chunk_ids=True, merge_fn=refuse)
    assert calls == [(k_rows, 0.1, 6, 0, 0)] and all(torch.equal(a, b) for a, b in zip(got, want))
    marker = (torch.zeros(1, W), torch.zeros(1, dtype=torch.long), torch.zeros(1, dtype=torch.long))
    assert parallel.merge_scene(blocks, k_rows, orc.nms, 0.1, merge_fn=lambda *a: marker) == marker[:2]
    assert parallel.merge_scene(blocks, k_rows, orc.nms, 0.1, with_chunk_ids=True, merge_fn=lambda *a: marker) == marker
    assert want[0].shape[0] == 8 and want[2].tolist() == sorted(want[2].tolist(), key=lambda c: 0) and set(want[2].tolist()) == {0, 2}


def test_merge_scene_real_record_with_minus_inf_score_is_not_dropped():
    """a valid record whose score is -inf used to tie with the padding rows (sort key -inf) and could lose its place
    to a padding row with a smaller flat index; validity now ranks before the score, as in the fused kernel's compaction."""
    import torch
    from sis3d import parallel
    W, k = parallel.RECORD_WIDTH, 4
    blocks = torch.zeros(2, parallel.block_floats(k))
    rows0 = torch.zeros(k, W)
    rows0[0, :6] = torch.tensor([0., 0., 0., 4., 4., 4.])
    rows0[0, 6] = 0.5
    blocks[0, 0], blocks[0, 1:] = 1, rows0.reshape(-1)          # chunk 0: one record, three padding rows (flat 1..3)
    rows1 = torch.zeros(k, W)
    rows1[0, :6] = torch.tensor([50., 0., 0., 54., 4., 4.])
    rows1[0, 6] = float("-inf")                                 # a real record with score -inf at flat index 4
    rows1[1, :6] = torch.tensor([70., 0., 0., 74., 4., 4.])
    rows1[1, 6] = 0.25
    blocks[1, 0], blocks[1, 1:] = 2, rows1.reshape(-1)
    recs, keep, cids = parallel.merge_scene(blocks, k, lambda b, th: torch.arange(b.shape[0]), 0.1, with_chunk_ids=True)
    assert recs.shape[0] == 3 and cids.tolist() == [0, 1, 1]
    assert recs[:, 0].tolist() == [0.0, 70.0, 50.0]             # 0.5, 0.25, then the -inf record -- not a zero padding row
