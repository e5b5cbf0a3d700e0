"""DocumentAnalyzer against the CPU oracle, free-running and end to end (tools/e2e_oracle_eval.py): the product's `serve`
on one side, oracle.pipeline.analyze - its own detector map, its own boxes, crops, layout, table crops, cell grids and
aggregation (oracle/hostlogic.py, pinned against the reference's functions) - on the other; nothing is handed across.  Two
pages whose calibrated heads find tables with rows, columns and cells; default arithmetic (fp16 planes) and exact fp32.
Every discrete leaf must be equal, or the page's difference must be traced to a tie in a continuous output that lies
within twice the north star's 1e-3 (document_analyzer.py:487-601, 622-678; reading_order.py:201)."""
import importlib.util
import os

import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.slow(order=2)]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_pages_with_tables_free_running_against_the_oracle(dev):
    spec = importlib.util.spec_from_file_location("e2e_oracle_eval", os.path.join(ROOT, "tools", "e2e_oracle_eval.py"))
    ev = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ev)
    result = ev.evaluate(6, only=[0, 5], log=lambda s: print(s))
    assert result["totals"]["tables"] >= 2 and result["totals"]["cells"] >= 10 and result["totals"]["words"] >= 300
    for mode, r in result["modes"].items():
        assert r["stages"]["words"]["leaves"] > 3000 and r["stages"]["cells"]["leaves"] > 50
        bad = {k: v for k, v in r["verdicts"].items() if k != "borderline"}
        assert not bad, (mode, [p[mode] for p in result["per_page"] if p[mode].get("verdict") not in (None, "borderline")])
    print({m: (r["stages"], r["verdicts"]) for m, r in result["modes"].items()})
