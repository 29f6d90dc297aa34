"""The driver parses bench.py's LAST stdout line; round 5's line had grown to 26.7 KB and came back unparsed (BENCH_r05.parsed = null).
driver_line() is the compaction: fed round 5's full line (profiles/r05c_bench_default.json) it must fit LINE_LIMIT and keep every key the
contract names (metric / value / config.workload / roofline / cpu_baseline)."""
import io
import json
import os
import sys
from contextlib import redirect_stdout

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


def _full():
    return json.load(open(os.path.join(ROOT, "profiles", "r05c_bench_default.json")))


def test_driver_line_fits_and_keeps_the_contract_keys():
    full = _full()
    assert len(json.dumps(full)) > 20_000                     # the line that was not parsed
    line = bench.driver_line(full)
    text = json.dumps(line)
    assert len(text) <= bench.LINE_LIMIT <= 4000
    assert "\n" not in text
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["config"]["workload"].startswith("synthetic 10000000x1000000")
    assert "model" not in line["config"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "dominant_kernel", "algorithmic_bytes_per_example"):
        assert k in line["roofline"], k
    assert abs(line["roofline"]["frac"] - line["roofline"]["achieved"] / line["roofline"]["peak"]) < 1e-4
    for k in ("value", "unit", "cores", "kind", "sample", "cpu_model"):
        assert k in line["cpu_baseline"], k
    assert abs(line["value"] - full["value"]) / full["value"] < 1e-5
    assert abs(line["ms_per_step"] - full["ms_per_step"]) / full["ms_per_step"] < 1e-5
    oc = line["other_configs"]
    assert set(oc) == set(full["other_configs"])
    for name, e in oc.items():
        assert set(e) <= {"value", "ms_per_step", "frac", "frac_basis", "effective_frac", "traffic_ratio", "cpu_value"}, name
        assert e["value"] > 0 and 0 < e["frac"] <= 1.0


def test_driver_line_sheds_before_it_overflows():
    full = _full()
    full["other_configs"] = {f"side_{i}": dict(v) for i in range(12) for v in full["other_configs"].values()}   # 12 x 7 summary entries: past the budget
    line = bench.driver_line(full)
    assert len(json.dumps(line)) <= bench.LINE_LIMIT
    assert "roofline" in line and "cpu_baseline" in line


def test_emit_puts_one_line_on_stdout_and_the_details_on_stderr(tmp_path, monkeypatch):
    from contextlib import redirect_stderr
    monkeypatch.setenv("FMX_BENCH_DETAILS_DIR", str(tmp_path))
    monkeypatch.delenv("FMX_BENCH_DETAILS", raising=False)
    out, err = io.StringIO(), io.StringIO()
    with redirect_stdout(out), redirect_stderr(err):
        bench.emit(_full())
    # stdout: the ONE line, whole, parseable as it stands -- however the driver reads it
    text = out.getvalue()
    assert text.count("\n") == 1 and len(text) <= bench.LINE_LIMIT + 1
    last = json.loads(text)
    assert last["metric"] and last["roofline"]["frac"] > 0 and last["cpu_baseline"]["value"] > 0
    # stderr: the details, line by line, none longer than the driver's own line may be; they add up to the full object, which is also in the file
    lines = err.getvalue().strip().split("\n")
    assert len(lines) > 10 and all(ln.startswith("DETAILS ") for ln in lines) and max(len(ln) for ln in lines) <= bench.LINE_LIMIT
    full = json.load(open(tmp_path / "bench_details.json"))
    assert bench.details_from_lines(lines) == full == _full()
    # FMX_BENCH_DETAILS=stdout: the details ahead of the line on stdout, the line still last; =off: the line alone
    monkeypatch.setenv("FMX_BENCH_DETAILS", "stdout")
    out = io.StringIO()
    with redirect_stdout(out):
        bench.emit(_full())
    ls = out.getvalue().strip().split("\n")
    assert json.loads(ls[-1])["value"] == last["value"] and all(ln.startswith("DETAILS ") for ln in ls[:-1])
