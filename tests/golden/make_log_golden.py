"""BUILD-CONTAINER ONLY: feeds log lines produced by THIS build's trainer formatters to the reference's
own scraper (release_scripts/log2csv.py:28-107 extract_final_metrics_from_log, imported from
/root/reference) and commits lines + what the reference parsed from them (tests/golden/log_contract.json).

    python tests/golden/make_log_golden.py
"""
import importlib.util
import io
import json
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def sample_log():
    """The four kinds of lines log2csv.py reads, written by clm_gs_amd.trainer's own formatters."""
    from types import SimpleNamespace

    import torch

    from clm_gs_amd import trainer
    buf = io.StringIO()
    buf.write("iteration[1,5) loss: 0.123456 0.234567 0.345678 0.456789 image: ['cam_00001', 'cam_00002', 'cam_00003', 'cam_00004']"
              " sparsity: 0.1000 0.1100 0.1200 0.1300\n")
    buf.write("[ITER {}] Evaluating {}: L1 {} PSNR {}\n".format(29997, "train", 0.013125141151249409, 33.42594528198242))
    buf.write("[ITER {}] Evaluating {}: L1 {} PSNR {}\n".format(29997, "test", 0.015133645385503769, 32.101051330566406))
    g = SimpleNamespace(get_xyz=torch.zeros(1215377, 1), parameters_buffer=torch.zeros(3, 48))
    keep = (torch.cuda.memory_allocated, torch.cuda.max_memory_allocated)
    torch.cuda.memory_allocated = lambda *a: int(1.25 * 2 ** 30)  # fixed figures: the same text with or without a GPU
    torch.cuda.max_memory_allocated = lambda *a: int(1.75 * 2 ** 30)
    try:
        line = trainer.memory_line(29997, 4, g)
    finally:
        torch.cuda.memory_allocated, torch.cuda.max_memory_allocated = keep
    buf.write(line)
    t = trainer.End2endTimer()
    t.total_time = 351.06
    t.print_time(buf, 30001)
    return buf.getvalue()


def main():
    text = sample_log()
    spec = importlib.util.spec_from_file_location("ref_log2csv", "/root/reference/release_scripts/log2csv.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    with tempfile.NamedTemporaryFile("w", suffix=".log", delete=False) as f:
        f.write(text)
    metrics = mod.extract_final_metrics_from_log(f.name)
    os.unlink(f.name)
    json.dump({"log": text, "parsed_by_reference": metrics}, open(os.path.join(HERE, "log_contract.json"), "w"), indent=1)
    print(metrics)


if __name__ == "__main__":
    main()
