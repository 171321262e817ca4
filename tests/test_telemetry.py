"""CPU: clm_gs_amd/telemetry.py (clocks / power / allocation telemetry beside the bench figures) -- the parsers, and that
every entry point degrades to an {"error": ...} record instead of raising where there is no device (this container)."""
import os

from clm_gs_amd import telemetry as T


def test_dpm_level_parser():
    assert T._dpm_current_mhz("0: 132Mhz\n1: 2100Mhz *") == 2100
    assert T._dpm_current_mhz("0: 500Mhz *\n1: 2400Mhz") == 500
    assert T._dpm_current_mhz("0: 1250Mhz") == 1250          # a single level without a marker
    assert T._dpm_current_mhz("") is None and T._dpm_current_mhz(None) is None


def test_sysfs_sample_from_a_fake_device_tree(tmp_path):
    d = tmp_path / "device"
    h = d / "hwmon" / "hwmon3"
    h.mkdir(parents=True)
    (d / "pp_dpm_sclk").write_text("0: 132Mhz\n1: 2297Mhz *\n")
    (d / "pp_dpm_mclk").write_text("0: 900Mhz\n1: 2000Mhz *\n")
    (d / "pp_dpm_fclk").write_text("0: 1250Mhz *\n")
    (h / "power1_average").write_text("983000000\n")
    (h / "power1_cap").write_text("1400000000\n")
    (h / "temp1_input").write_text("51000\n")
    (h / "temp1_label").write_text("junction\n")
    (h / "temp2_input").write_text("54000\n")
    (h / "temp2_label").write_text("mem\n")
    (d / "current_compute_partition").write_text("SPX\n")
    (d / "current_memory_partition").write_text("NPS1\n")
    (d / "mem_info_vram_used").write_text(str(42 * 2 ** 30))
    s = T._sample_sysfs(str(d), full=True)
    assert s["sclk_mhz"] == 2297 and s["mclk_mhz"] == 2000 and s["fclk_mhz"] == 1250
    assert s["power_w"] == 983.0 and s["power_cap_w"] == 1400.0
    assert s["temp_c"] == {"junction": 51.0, "mem": 54.0}
    assert s["compute_partition"] == "SPX" and s["memory_partition"] == "NPS1" and s["vram_used_gb"] == 42.0
    quick = T._sample_sysfs(str(d))
    assert set(quick) == {"sclk_mhz", "power_w"}


def test_snapshot_and_sampler_never_raise_without_a_device():
    s = T.snapshot()
    assert isinstance(s, dict) and "source" in s
    with T.Sampler(hz=50) as smp:
        pass
    out = smp.summary()
    assert set(out) >= {"source", "start", "end", "sclk_mhz", "power_w"}
    assert T.tensor_alloc_info({}) == {} and T.tensor_alloc_info({"x": None}) == {}
