"""bench.py's host logic that needs no GPU: the committed PMC counters are reported only for
the kernel they were collected on."""
import json
import os
import sys

from conftest import ROOT

sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_source_hash_changes_with_the_device_sources(tmp_path, monkeypatch):
    h = bench.kernel_source_hash()
    assert len(h) == 40 and h == bench.kernel_source_hash()
    src = tmp_path / "path-tracer_amd" / "csrc"
    src.mkdir(parents=True)
    (src / "a.hpp").write_text("x")
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    h1 = bench.kernel_source_hash()
    (src / "a.hpp").write_text("y")
    assert bench.kernel_source_hash() != h1 != h


def test_counters_are_dropped_when_stale(tmp_path, monkeypatch):
    (tmp_path / "profiles").mkdir()
    (tmp_path / "path-tracer_amd" / "csrc").mkdir(parents=True)
    (tmp_path / "path-tracer_amd" / "csrc" / "k.hip").write_text("kernel")
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    pmc, why = bench.pmc_summary(40.0)
    assert pmc == {} and "missing" in why
    rec = {"hbm_bytes_per_launch": 1.0, "kernel_ms": 40.0, "source_sha1": bench.kernel_source_hash()}
    path = tmp_path / "profiles" / "pmc_traffic.json"
    path.write_text(json.dumps(rec))
    pmc, why = bench.pmc_summary(41.0)            # within 5 %
    assert pmc["hbm_bytes_per_launch"] == 1.0 and why is None
    pmc, why = bench.pmc_summary(43.0)            # the kernel runs 7.5 % slower than when profiled
    assert pmc == {} and "ms" in why
    (tmp_path / "path-tracer_amd" / "csrc" / "k.hip").write_text("another kernel")
    pmc, why = bench.pmc_summary(40.0)
    assert pmc == {} and "other device sources" in why
    path.write_text(json.dumps({"hbm_bytes_per_launch": 1.0}))      # a summary from before the stamp existed
    pmc, why = bench.pmc_summary(40.0)
    assert pmc == {} and why


def test_the_committed_summary_carries_the_stamp():
    with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
        rec = json.load(f)
    assert "source_sha1" in rec and rec.get("kernel_ms", 0) > 0
