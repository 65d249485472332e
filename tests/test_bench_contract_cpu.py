"""bench.py's launcher contract and its counter-profile staleness guard, without a GPU."""
import importlib.util
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("ppv_bench", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_gpus_flag_is_checked_against_the_visible_devices_and_the_launched_world():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "PPV_FORCE_DEVICE0")}
    env["HIP_VISIBLE_DEVICES"] = ""                                   # no device visible: --gpus 2 cannot be honoured
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 2 and "refusing" in r.stderr and "{" not in r.stdout
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], env=dict(env, WORLD_SIZE="4"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and "WORLD_SIZE=4" in r.stderr and "{" not in r.stdout


def test_counter_fields_are_null_when_the_profile_was_taken_on_other_kernel_sources(tmp_path, monkeypatch):
    b = _bench()
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from csrc_hash import csrc_hash
    good = tmp_path / "r99z_pmc_traffic.json"
    good.write_text(json.dumps({"csrc_sha16": csrc_hash(), "total_fetch_GB_per_step": 60.0, "total_write_GB_per_step": 30.0, "per_kernel": {}}))
    bad = tmp_path / "r99y_pmc_traffic.json"
    bad.write_text(json.dumps({"csrc_sha16": "0" * 16, "total_fetch_GB_per_step": 60.0, "total_write_GB_per_step": 30.0, "per_kernel": {}}))
    legacy = tmp_path / "r99x_pmc_traffic.json"
    legacy.write_text(json.dumps({"total_fetch_GB_per_step": 60.0, "total_write_GB_per_step": 30.0, "per_kernel": {}}))
    d, stale = b._load_profile(str(good))
    assert d is not None and not stale
    for f in (bad, legacy):
        d, stale = b._load_profile(str(f))
        assert d is None and stale
    monkeypatch.setattr(b, "_latest_profile", lambda suffix: str(bad))
    h = b._step_hbm(0.02)
    assert h["GB_per_step_pmc"] is None and h["stale"] is True
    monkeypatch.setattr(b, "_latest_profile", lambda suffix: str(good))
    h = b._step_hbm(0.02)
    assert h["GB_per_step_pmc"] == 90.0 and "stale" not in h


def test_hash_changes_with_the_sources(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from csrc_hash import csrc_hash
    d = tmp_path / "privacy-preserving-vision_amd" / "csrc"
    d.mkdir(parents=True)
    (d / "a.hip").write_text("x")
    h1 = csrc_hash(str(tmp_path))
    (d / "a.hip").write_text("y")
    assert csrc_hash(str(tmp_path)) != h1
