"""CellDetector on the MI355X (reference table_cell_detector.py:195-524): RT-DETRv2 at 960 x 960 / 1500 queries behind
the module API.  The network is pinned by the reference-class golden in tests/test_rtdetr_gpu.py; here the module is
checked end to end against the ORACLE chain on the same page: PIL-resized crops -> oracle forward -> the (pinned)
post-processing."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_cell_detector_matches_oracle_chain(dev):
    from oracle.rtdetr import rtdetr_forward
    from tests.test_rtdetr_gpu import assert_same_detections
    from yomitoku_amd.schemas import Element, TableDetectorSchema
    from yomitoku_amd.table_cell_detector import CellDetector
    from yomitoku_amd.utils.synth import synthetic_page_with_truth
    from yomitoku_amd.utils.synth_rtdetr import rtdetr_state_dict

    img, _, tables, _ = synthetic_page_with_truth(3, 1000, 1400)
    tables = (tables + [[40, 60, 900, 500]])[:2]
    elems = [Element(id=None, box=b, score=1.0, role=None, contents=None) for b in tables]
    det = CellDetector(from_pretrained=False, device="cuda:0")
    assert det._cfg.data.img_size == [960, 960] and det._cfg.RTDETRTransformerv2.num_queries == 1500
    sd = rtdetr_state_dict(1243, num_classes=6, eval_size=(960, 960), enc_score_gain=12.0, score_bias=-3.0, score_gain=2.0)
    det.model.load_state_dict(sd)
    batch, metas = det.preprocess(img, elems)
    assert batch.shape == (len(elems), 3, 960, 960)
    # crops: the installed Pillow is the reference's own resize (cell detector :318-337)
    from PIL import Image

    for k, box in enumerate(tables):
        x1, y1, x2, y2 = box
        crop = Image.fromarray(np.ascontiguousarray(img[y1:y2, x1:x2, ::-1])).resize((960, 960), Image.BILINEAR)
        want = torch.from_numpy(np.asarray(crop)).permute(2, 0, 1).float() / 255.0
        assert torch.equal(batch[k].cpu(), want)
    preds = det.model(batch)
    ref = rtdetr_forward(sd, batch.cpu(), num_queries=1500)
    lg, bx = preds["pred_logits"].cpu().numpy(), preds["pred_boxes"].cpu().numpy()
    assert_same_detections(lg, bx, ref["pred_logits"].numpy(), ref["pred_boxes"].numpy())
    out = det(img, elems)
    assert all(isinstance(t, TableDetectorSchema) for t in out) and len(out) >= 1
    # the module's cells == the post-processing applied to the oracle's predictions, as a SET per role: detections come
    # out in score order, and two scores a few ulp apart may swap between two implementations (boxes within a pixel:
    # the cast to int may fall either side when a coordinate sits on an integer)
    for k, (table, data) in enumerate(zip(out, metas)):
        cells, kv, grid = det.postprocess({"pred_logits": ref["pred_logits"][k : k + 1].numpy(), "pred_boxes": ref["pred_boxes"][k : k + 1].numpy()},
                                          data, elems[k].box)
        want = sorted((c.role, *c.box) for c in cells)
        got = sorted((c.role, *c.box) for c in table.cells)
        assert [w[0] for w in want] == [g[0] for g in got]
        assert np.abs(np.array([w[1:] for w in want]) - np.array([g[1:] for g in got])).max() <= 1
        assert [c.id for c in table.cells] == [f"c{i}" for i in range(len(table.cells))]
        assert len(kv) == len(table.kv_regions) and len(grid) == len(table.grid_regions)
    print("cells per table", [len(t.cells) for t in out])
