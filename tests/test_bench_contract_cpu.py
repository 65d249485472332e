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


def test_live_counter_passes_take_precedence_and_fall_back_cleanly(monkeypatch):
    """bench.py measures roofline.traffic / step_hbm in its own run (rocprofv3 child passes, round 5); where rocprofv3 is missing the
    call leaves nothing behind and the committed profiles are used; a live result overrides them and is labelled."""
    b = _bench()
    monkeypatch.setattr("shutil.which", lambda name: None)
    b._LIVE.clear()
    b.live_counter_passes()
    assert b._LIVE == {}
    b._LIVE["pmc"] = {"total_fetch_GB_per_step": 60.0, "total_write_GB_per_step": 32.0, "per_kernel": {}}
    b._LIVE["mfma"] = {"per_kernel": {}}
    h = b._step_hbm(0.0224)
    assert h["GB_per_step_pmc"] == 92.0 and h["pmc_source"].startswith("live") and abs(h["TB_per_s"] - 4.11) < 0.01
    b._LIVE.clear()


def test_ic_transform_length_is_the_next_native_length():
    """Patch sizes off the 128 / 256 grid run on the next 256 / 512 / 1024-point transform (>= 2 P: the linear convolution of two P x P
    supports fits); above 512 there is no native transform."""
    import pytest
    sys.path.insert(0, ROOT)
    import ppv_amd  # noqa: F401
    import ppv_amd.fftconv as fc
    assert [fc.ic_transform_length(p) for p in (32, 128, 130, 256, 258, 368, 512)] == [256, 256, 512, 512, 1024, 1024, 1024]
    with pytest.raises(ValueError):
        fc.ic_transform_length(514)


def test_headline_is_the_dense_surface_as_a_median_of_three_windows():
    """VERDICT r5 task 4: `value` is the dense module surface (models.py:39-41), `value_lazy_consumer` the extra key; both the median of
    three K-step windows, the windows printed, the GC state recorded."""
    b = _bench()
    calls = []
    wins, enq = b.timed_windows(lambda: calls.append(1), 5, 3, 1, None, sync=lambda: None)
    assert len(calls) == 15 and len(wins) == 3 and len(enq) == 3 and all(w >= e for w, e in zip(wins, enq))
    f = b.surface_fields(1, 128, 20, [0.46, 0.44, 0.50], [0.42, 0.43, 0.41], True, False)
    for key in ("windows_ms_per_step", "timing", "gc", "surface", "value_dense_surface", "ms_per_step_dense_surface", "value_lazy_consumer",
                "ms_per_step_lazy_consumer", "windows_ms_per_step_other_surface"):
        assert key in f
    assert f["windows_ms_per_step"] == [23.0, 22.0, 25.0] and f["ms_per_step_dense_surface"] == 23.0     # the median window
    assert f["value_dense_surface"] == round(128 * 20 / 0.46, 1) and f["value_lazy_consumer"] == round(128 * 20 / 0.42, 1)
    assert f["surface"].startswith("dense") and "median of 3 windows of exactly 20 steps" in f["timing"] and "frozen" in f["gc"]
    lz = b.surface_fields(1, 128, 20, [0.42], None, False, False)                                       # --lazy diagnosis run
    assert lz["value_dense_surface"] is None and lz["value_lazy_consumer"] == round(128 * 20 / 0.42, 1)
    dec = b.surface_fields(1, 128, 20, [0.5], None, False, True)                                         # config 3: neither surface key
    assert dec["value_dense_surface"] is None and dec["value_lazy_consumer"] is None

    class Enc:
        lazy_output = True
    e = Enc()
    with b.surface_mode(e, True):
        assert e.lazy_output is False and b._DENSE_HEAD[0] is True
    assert e.lazy_output is True and b._DENSE_HEAD[0] is False
