"""Device telemetry beside the throughput figures: clocks, socket power, power cap, temperatures, partition modes, memory.

The reference prints memory lines at every densification (utils/general_utils.py:216-240 check_memory_usage:
memory_allocated / max_memory_allocated / reserved); on an MI355X the figure that moves a VALU-bound kernel from one
run to the next is the shader clock under the socket power cap, and the figure that moves a gather is how the tables
were allocated -- so this module adds those to the same place.

Sources, in this order (the first that answers is used for the whole process):
  1. amdgpu sysfs of the device torch runs on (/sys/class/drm/card*/device: pp_dpm_sclk / pp_dpm_mclk / pp_dpm_fclk,
     hwmon power1_average | power1_input, power1_cap, temp*_input + temp*_label, current_compute_partition,
     current_memory_partition, mem_info_vram_used) -- plain file reads, ~0.2 ms per sample, usable from a sampling thread;
  2. `rocm-smi --json` (a subprocess, ~1 s per sample): only `snapshot()` at the ends of a region.

Nothing here touches the hot path: `Sampler` runs on a host thread at 10 Hz and only reads files.
"""
import glob
import json
import os
import re
import subprocess
import threading
import time

_SYSFS = None       # resolved device directory, or False
_SOURCE = None


def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def _device_dir(index=0):
    """sysfs directory of the `index`-th amdgpu device that exposes pp_dpm_sclk (render order = HIP order on one-GPU
    boxes; on multi-GPU nodes the PCI bus id of torch's device is matched when torch is importable)."""
    global _SYSFS
    if _SYSFS is not None:
        return _SYSFS or None
    cands = sorted(d for d in glob.glob("/sys/class/drm/card[0-9]*/device") if os.path.exists(os.path.join(d, "pp_dpm_sclk")))
    pick = None
    try:
        import torch
        if torch.cuda.is_available():
            p = torch.cuda.get_device_properties(torch.cuda.current_device())
            bus = "%04x:%02x:%02x" % (getattr(p, "pci_domain_id", 0), getattr(p, "pci_bus_id", -1), getattr(p, "pci_device_id", 0))
            for d in cands:
                if bus in os.path.realpath(d):
                    pick = d
    except Exception:
        pass
    if pick is None and cands:
        pick = cands[min(index, len(cands) - 1)]
    _SYSFS = pick or False
    return pick


def _dpm_current_mhz(text):
    """pp_dpm_* lists the levels, the current one carries '*': '0: 132Mhz\\n1: 2100Mhz *'."""
    if not text:
        return None
    cur = None
    for line in text.splitlines():
        m = re.search(r"(\d+)\s*[Mm][Hh]z", line)
        if m and "*" in line:
            cur = int(m.group(1))
    if cur is None:
        m = re.findall(r"(\d+)\s*[Mm][Hh]z", text)
        cur = int(m[-1]) if m else None
    return cur


def _hwmon(d):
    h = glob.glob(os.path.join(d, "hwmon", "hwmon*"))
    return h[0] if h else None


def _sample_sysfs(d, full=False):
    out = {"sclk_mhz": _dpm_current_mhz(_read(os.path.join(d, "pp_dpm_sclk")))}
    h = _hwmon(d)
    if h:
        p = _read(os.path.join(h, "power1_average")) or _read(os.path.join(h, "power1_input"))
        out["power_w"] = round(int(p) / 1e6, 1) if p and p.isdigit() else None
    if full:
        out["mclk_mhz"] = _dpm_current_mhz(_read(os.path.join(d, "pp_dpm_mclk")))
        out["fclk_mhz"] = _dpm_current_mhz(_read(os.path.join(d, "pp_dpm_fclk")))
        if h:
            c = _read(os.path.join(h, "power1_cap"))
            out["power_cap_w"] = round(int(c) / 1e6, 1) if c and c.isdigit() else None
            temps = {}
            for f in sorted(glob.glob(os.path.join(h, "temp*_input"))):
                lab = _read(f.replace("_input", "_label")) or os.path.basename(f)
                v = _read(f)
                if v and v.lstrip("-").isdigit():
                    temps[lab] = round(int(v) / 1000.0, 1)
            out["temp_c"] = temps
        out["compute_partition"] = _read(os.path.join(d, "current_compute_partition"))
        out["memory_partition"] = _read(os.path.join(d, "current_memory_partition"))
        v = _read(os.path.join(d, "mem_info_vram_used"))
        out["vram_used_gb"] = round(int(v) / 2 ** 30, 2) if v and v.isdigit() else None
        out["perf_level"] = _read(os.path.join(d, "power_dpm_force_performance_level"))
    return out


def _sample_rocm_smi():
    try:
        r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showmaxpower", "--showtemp", "--showperflevel",
                            "--showcomputepartition", "--showmemorypartition", "--json"],
                           capture_output=True, text=True, timeout=20)
        txt = r.stdout[r.stdout.index("{"):]
        card = next(iter(json.loads(txt).values()))
    except Exception as e:  # noqa: BLE001
        return {"error": f"rocm-smi: {e}"}

    def mhz(key):
        for k, v in card.items():
            if k.startswith(key):
                m = re.search(r"(\d+)", str(v))
                return int(m.group(1)) if m else None
        return None

    def num(sub):
        for k, v in card.items():
            if sub in k:
                try:
                    return float(v)
                except (TypeError, ValueError):
                    pass
        return None
    return {"sclk_mhz": mhz("sclk clock speed"), "mclk_mhz": mhz("mclk clock speed"), "fclk_mhz": mhz("fclk clock speed"),
            "power_w": num("Package Power"), "power_cap_w": num("Max Graphics Package Power"),
            "temp_c": {k: v for k, v in card.items() if "Temperature" in k},
            "compute_partition": next((v for k, v in card.items() if "Compute Partition" in k), None),
            "memory_partition": next((v for k, v in card.items() if "Memory Partition" in k), None),
            "perf_level": next((v for k, v in card.items() if "Performance Level" in k), None)}


def source():
    global _SOURCE
    if _SOURCE is None:
        _SOURCE = "sysfs" if _device_dir() else "rocm-smi"
    return _SOURCE


def snapshot():
    """One full reading (clocks, power + cap, temperatures, partition modes, memory) -> dict; never raises."""
    try:
        d = _device_dir()
        out = _sample_sysfs(d, full=True) if d else _sample_rocm_smi()
    except Exception as e:  # noqa: BLE001
        out = {"error": repr(e)}
    out["source"] = source()
    try:
        import torch
        if torch.cuda.is_available():
            free, total = torch.cuda.mem_get_info()
            out["hbm_free_gb"], out["hbm_total_gb"] = round(free / 2 ** 30, 2), round(total / 2 ** 30, 2)
    except Exception:
        pass
    return out


class Sampler:
    """Samples sclk + socket power on a host thread while a region runs (sysfs only; with rocm-smi as the source the
    region gets the two end snapshots).  `with Sampler() as s: ...; s.summary()` -> start / end snapshots and
    min / mean / max over the region."""

    def __init__(self, hz=10.0):
        self.period = 1.0 / hz
        self.samples = []
        self._stop = threading.Event()
        self._th = None
        self.start_state = self.end_state = None

    def __enter__(self):
        self.start_state = snapshot()
        d = _device_dir()
        if d:
            def run():
                while not self._stop.wait(self.period):
                    try:
                        s = _sample_sysfs(d)
                        s["t"] = time.perf_counter()
                        self.samples.append(s)
                    except Exception:
                        return
            self._th = threading.Thread(target=run, name="clmgs-telemetry", daemon=True)
            self._th.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        if self._th is not None:
            self._th.join()
        self.end_state = snapshot()
        return False

    def summary(self):
        def stat(key):
            v = [s[key] for s in self.samples if s.get(key) is not None]
            if not v:
                return None
            return {"min": min(v), "mean": round(sum(v) / len(v), 1), "max": max(v), "n": len(v)}
        return {"source": source(), "start": self.start_state, "end": self.end_state,
                "sclk_mhz": stat("sclk_mhz"), "power_w": stat("power_w")}


def tensor_alloc_info(named_tensors):
    """How the caching allocator placed each (large) tensor: its bytes, its address alignment, and the segment
    (one hipMalloc) it lives in -- segment size and the tensor's offset inside it.  A table that sits alone in a
    2 MB-aligned segment of its own was one hipMalloc; one at an offset inside a larger segment shares it."""
    import torch
    segs = []
    try:
        for s in torch.cuda.memory_snapshot():
            segs.append((int(s["address"]), int(s["total_size"]), s.get("segment_type", "?")))
    except Exception:
        pass
    out = {}
    for name, t in named_tensors.items():
        if t is None or not getattr(t, "is_cuda", False):
            continue
        ptr = t.data_ptr()
        info = {"bytes": t.numel() * t.element_size(), "align_2mb": ptr % (2 << 20) == 0, "align_64kb": ptr % (64 << 10) == 0}
        for a, sz, kind in segs:
            if a <= ptr < a + sz:
                info.update(segment_bytes=sz, offset_in_segment=ptr - a, segment_type=kind,
                            own_segment=bool(ptr == a and sz - info["bytes"] < (2 << 20)))
                break
        out[name] = info
    return out
