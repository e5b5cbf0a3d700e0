"""Integer geometry and batching rules of the detector / recogniser front ends against answers of the REFERENCE's own
functions (oracle/pin_against_reference.py geometry -> tests/golden/geometry.json): exact."""
import json
import os
from types import SimpleNamespace

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden", "geometry.json")


@pytest.fixture(scope="module")
def gold():
    with open(GOLD) as f:
        return json.load(f)


def test_detector_resize_sizes(gold):
    from yomitoku_amd.imaging import resize_shortest_edge_dims

    assert len(gold["resize"]) >= 300
    for c in gold["resize"]:
        assert list(resize_shortest_edge_dims(c["h"], c["w"], 1280, 1600)) == c["out"], (c["h"], c["w"])


def test_crop_resize_sizes_and_quad_validation(gold):
    """calc_resize_without_padding / validate_quads as the crop planner applies them."""
    from yomitoku_amd import imaging

    for c in gold["pad"]:
        h, w = c["h"], c["w"]
        plan = imaging.plan_crops((h + 4, w + 4), [[[0, 0], [w, 0], [w, h], [0, h]]], (32, 800), False)[0]
        rot = h > 2 * w
        want_h, want_w = c["out"]
        if not rot:  # the pinned function sees the un-rotated h x w image
            assert (plan.desc.nh, plan.desc.nw) == (want_h, want_w), (h, w)
    for c in gold["quads"]:
        assert imaging.validate_quad((300, 400), c["quad"]) == c["valid"]
        if not c["valid"]:
            assert imaging.plan_crops((300, 400), [c["quad"]])[0] is None


def test_source_levels(gold):
    from yomitoku_amd.imaging import source_levels

    for c in gold["levels"]:
        assert source_levels(c["quads"], 32).tolist() == c["levels"]


def test_mini_batch_compositions(gold):
    """TextRecognizer._make_mini_batch: width-budget batching, max batch size, fixed batch size, bucketing order."""
    from yomitoku_amd.text_recognizer import TextRecognizer

    seen_budget = seen_fixed = 0
    for c in gold["batches"]:
        data = SimpleNamespace(batch_size=c["batch_size"], width_budget=c["width_budget"], max_batch_size=c["max_batch_size"],
                               img_size=[32, 800])
        fake = SimpleNamespace(_cfg=SimpleNamespace(data=data), dynamic_width=c["dynamic"])
        plans = [SimpleNamespace(index=i, canvas_width=w) for i, w in enumerate(c["widths"])]

        class DS:
            def __init__(self, p):
                self.plans = p

            def __len__(self):
                return len(self.plans)

        batches = TextRecognizer._make_mini_batch(fake, DS(plans), c["order"])
        got = [{"members": [p.index for p in b], "width": max(p.canvas_width for p in b) if c["dynamic"] else 800} for b in batches]
        assert got == c["batches"]
        seen_budget += bool(c["dynamic"] and c["width_budget"])
        seen_fixed += not (c["dynamic"] and c["width_budget"])
    assert seen_budget >= 10 and seen_fixed >= 10


def test_forward_chunks_respect_the_line_and_token_row_bounds():
    """TextRecognizer._forward_chunks: consecutive mini-batches share a forward up to MAX_LINES_PER_FORWARD
    lines AND up to the token-row budget of the model's width; forwards come out about equal in size; an oversize mini-batch
    runs alone; the reservation bounds cover every forward the rule can form."""
    import random

    from yomitoku_amd.text_recognizer import TextRecognizer

    def fake(dim, patch, dynamic, width=800):
        cfg = SimpleNamespace(data=SimpleNamespace(img_size=[32, width]), encoder=SimpleNamespace(patch_size=list(patch), embed_dim=dim))
        f = SimpleNamespace(_cfg=cfg, dynamic_width=dynamic, MAX_LINES_PER_FORWARD=TextRecognizer.MAX_LINES_PER_FORWARD,
                            FORWARD_WORKSPACE_BYTES=TextRecognizer.FORWARD_WORKSPACE_BYTES)
        for name in ("_token_geometry", "_job_token_rows", "_forward_chunks", "_reserve_bounds"):
            setattr(f, name, getattr(TextRecognizer, name).__get__(f))
        return f

    rng = random.Random(5)
    for dim, patch, dynamic in ((192, (4, 8), True), (768, (8, 8), False), (512, (8, 8), True)):
        rec = fake(dim, patch, dynamic)
        full, budget = rec._token_geometry()
        assert full == (32 // patch[0]) * (800 // patch[1]) and budget == (16 << 30) // (52 * dim)
        lines, h, w = rec._reserve_bounds()
        assert lines == 2048 and h == 32 and w % patch[1] == 0 and w <= 800
        assert lines * (32 // patch[0]) * (w // patch[1]) <= budget or w == patch[1]
        for trial in range(30):
            jobs = []
            for _ in range(rng.randint(1, 90)):
                n = rng.randint(1, 128)
                jobs.append((None, [SimpleNamespace(canvas_width=rng.choice((64, 160, 320, 800))) for _ in range(n)]))
            chunks = rec._forward_chunks(jobs)
            assert chunks[0][0] == 0 and chunks[-1][1] == len(jobs) and all(a[1] == b[0] for a, b in zip(chunks, chunks[1:]))
            for lo, hi in chunks:
                nl = sum(len(p) for _, p in jobs[lo:hi])
                nr = sum(rec._job_token_rows(p) for _, p in jobs[lo:hi])
                assert hi > lo and (hi - lo == 1 or (nl <= 2048 and nr <= budget))
    # the default recogniser's fixed 800 px canvas: 16 pages x 100 lines no longer land in ONE forward of 1600 lines
    rec = fake(768, (8, 8), False)
    jobs = [(None, [SimpleNamespace(canvas_width=800)] * 100) for _ in range(16)]
    sizes = [sum(len(p) for _, p in jobs[lo:hi]) for lo, hi in rec._forward_chunks(jobs)]
    assert sizes == [800, 800]
    # the --lite recogniser keeps its 2048-line forwards (the line bound is the tighter one)
    rec = fake(192, (4, 8), True)
    jobs = [(None, [SimpleNamespace(canvas_width=800)] * 128) for _ in range(20)]
    assert [sum(len(p) for _, p in jobs[lo:hi]) for lo, hi in rec._forward_chunks(jobs)] == [1280, 1280]


def test_tokenizer_decode(gold):
    from yomitoku_amd.text_recognizer import ParseqTokenizer

    c = gold["tokenizer"]
    tok = ParseqTokenizer(c["charset"])
    probs = np.asarray(c["probs"], dtype=np.float32)
    texts, scores = tok.decode_stats(probs.argmax(-1), probs.max(-1))
    assert texts == c["texts"]
    assert np.allclose(scores, c["scores"], rtol=1e-6, atol=0)
    assert any(len(t) == probs.shape[1] for t in texts)  # the row that never reaches <eos>
