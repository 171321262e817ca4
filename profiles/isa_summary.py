"""Static resource / instruction-mix table of every kernel in clm_gs_amd/csrc (gfx950), from the compiler's own
assembly (`hipcc -S`, same flags as the Makefile).  No GPU needed:  python profiles/isa_summary.py > profiles/rNN_isa_summary.txt

Per kernel: VGPRs, SGPRs, spilled VGPRs, scratch bytes, static LDS, waves/SIMD those registers allow (512 VGPRs per
SIMD lane, allocation granule 8, at most 8 waves), static instruction counts by class (the whole kernel body, NOT
weighted by how often a loop runs -- a size / mix indicator, e.g. how little of the alpha-blend backward is cross-lane
traffic), and the `__launch_bounds__` occupancy the source asks for is visible as the VGPR budget the compiler kept."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "clm_gs_amd", "csrc")
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-munsafe-fp-atomics", "--cuda-device-only", "-S",
         "-I" + os.path.join(ROOT, "include")]
NO_SLP = {"rasterize.hip", "loss.hip"}  # as in the Makefile

CLASSES = [
    ("valu_fp", re.compile(r"^v_(mul|add|sub|fma|fmac|mad|max|min|rcp|rsq|sqrt|exp|log|cvt|med3|ldexp|frexp|floor|ceil|rndne|trunc|fract|pk_)\w*f(16|32|64)")),
    ("valu_trans", re.compile(r"^v_(rcp|rsq|sqrt|exp|log|sin|cos)_")),
    ("cross_lane", re.compile(r"(_dpp|^v_permlane|^ds_swizzle|^ds_bpermute|^ds_permute|^v_readlane|^v_readfirstlane|^v_writelane)")),
    ("mfma", re.compile(r"^v_mfma")),
    ("v_mov/cndmask", re.compile(r"^v_(mov|cndmask|accvgpr)")),
    ("valu_int/cmp", re.compile(r"^v_")),
    ("lds", re.compile(r"^ds_")),
    ("vmem_load", re.compile(r"^(global|buffer|flat|scratch)_load")),
    ("vmem_store", re.compile(r"^(global|buffer|flat|scratch)_(store|atomic)")),
    ("smem", re.compile(r"^s_(load|buffer_load)")),
    ("branch", re.compile(r"^s_(cbranch|branch|setpc|call)")),
    ("waitcnt/nop", re.compile(r"^s_(waitcnt|nop|barrier|sleep)")),
    ("salu", re.compile(r"^s_")),
]


def classify(op):
    if CLASSES[2][1].search(op):
        return "cross_lane"
    for name, rx in CLASSES:
        if name == "cross_lane":
            continue
        if rx.search(op):
            if name == "valu_fp" and CLASSES[1][1].search(op):
                return "valu_trans"
            return name
    return "other"


def waves_per_simd(vgprs):
    g = max(8, (vgprs + 7) // 8 * 8)
    return min(8, 512 // g)


def waves_lds(lds, wg):
    """Waves per SIMD the static LDS allows (160 KB per CU, 4 SIMDs) when the kernel is launched with its
    __launch_bounds__ block size (an upper bound for kernels launched with smaller blocks)."""
    if not lds:
        return 8
    blocks = (160 * 1024) // lds
    return min(8, blocks * max(1, wg // 64) // 4)


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True,
                             text=True, check=True).stdout.splitlines()
        return dict(zip(names, out))
    except Exception:
        return {n: n for n in names}


def main():
    rows = []
    with tempfile.TemporaryDirectory() as tmp:
        for src in sorted(f for f in os.listdir(CSRC) if f.endswith(".hip")):
            s_path = os.path.join(tmp, src + ".s")
            cmd = ["/opt/rocm/bin/hipcc"] + FLAGS + (["-fno-slp-vectorize"] if src in NO_SLP else []) + \
                  [os.path.join(CSRC, src), "-o", s_path]
            subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
            text = open(s_path).read()
            meta = {}
            # metadata block: one YAML map per kernel
            for blk in re.split(r"\n  - \.", text.split("amdhsa.kernels:")[-1]):
                nm = re.search(r"\.name:\s+(\S+)", blk) or re.search(r"^name:\s+(\S+)", blk, re.M)
                if not nm:
                    continue
                g = lambda k: int((re.search(r"\.?%s:\s+(\d+)" % k, blk) or [0, 0])[1])
                meta[nm.group(1)] = dict(vgpr=g("vgpr_count"), sgpr=g("sgpr_count"), spill=g("vgpr_spill_count"),
                                         scratch=g("private_segment_fixed_size"), lds=g("group_segment_fixed_size"),
                                         wg=g("max_flat_workgroup_size"))
            for name, md in meta.items():
                m = re.search(r"^%s:.*?\n(.*?)\n\s+s_endpgm" % re.escape(name), text, re.S | re.M)
                counts = {}
                n = 0
                if m:
                    for line in m.group(1).splitlines():
                        mm = re.match(r"^\s+([a-z][a-z0-9_]+)", line)
                        if not mm or line.lstrip().startswith((".", ";")):
                            continue
                        c = classify(mm.group(1))
                        counts[c] = counts.get(c, 0) + 1
                        n += 1
                rows.append((src, name, md, n, counts))
    dm = demangle([r[1] for r in rows])
    cols = ["valu_fp", "valu_trans", "valu_int/cmp", "v_mov/cndmask", "cross_lane", "mfma", "lds", "vmem_load", "vmem_store",
            "smem", "salu", "branch", "waitcnt/nop"]
    print("# static ISA summary, gfx950, hipcc -O3 (flags of clm_gs_amd/csrc/Makefile); counts are instructions in the "
          "kernel body, not weighted by trip counts")
    print("\t".join(["file", "kernel", "vgpr", "sgpr", "spill", "scratch_B", "lds_B", "max_wg", "waves/SIMD(vgpr)", "waves/SIMD(lds,max_wg)", "instr"] + cols))
    for src, name, md, n, counts in rows:
        short = re.sub(r"\(.*", "", dm[name]).replace("void ", "").replace("clmgs::", "").replace("HIP_vector_type", "vec")
        print("\t".join([src, short, str(md["vgpr"]), str(md["sgpr"]), str(md["spill"]), str(md["scratch"]),
                         str(md["lds"]), str(md["wg"]), str(waves_per_simd(md["vgpr"])), str(waves_lds(md["lds"], md["wg"])), str(n)] + [str(counts.get(c, 0)) for c in cols]))


if __name__ == "__main__":
    sys.exit(main())
