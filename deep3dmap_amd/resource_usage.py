"""Register / scratch / LDS budget of every kernel of libd3m_raster.so, as the compiler reports it
(`hipcc -Rpass-analysis=kernel-resource-usage`; cross-compiles without a GPU).  Used by the CPU test suite to keep
the hot kernels free of scratch spills and by tools_dev/ to print the table.

    python -m deep3dmap_amd.resource_usage [name-substring ...]
"""
import os
import re
import subprocess
import sys
import tempfile

from .build import CSRC, HIPCC_FLAGS, SOURCES

_FIELDS = {"sgprs": r"TotalSGPRs", "vgprs": r"VGPRs", "agprs": r"AGPRs", "scratch": r"ScratchSize \[bytes/lane\]",
           "occupancy": r"Occupancy \[waves/SIMD\]", "sgpr_spill": r"SGPRs Spill", "vgpr_spill": r"VGPRs Spill",
           "lds": r"LDS Size \[bytes/block\]"}


def _demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout
    return out.splitlines()


def kernel_resource_usage(extra_flags=()):
    """{demangled kernel name: {"vgprs", "sgprs", "agprs", "scratch", "occupancy", "vgpr_spill", "sgpr_spill", "lds"}}
    for the product build's flags (+ extra_flags)."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    with tempfile.TemporaryDirectory() as tmp:
        cmd = ([hipcc] + HIPCC_FLAGS + list(extra_flags) + ["-Rpass-analysis=kernel-resource-usage"] +
               [os.path.join(CSRC, s) for s in SOURCES] + ["-o", os.path.join(tmp, "lib.so")])
        err = subprocess.run(cmd, capture_output=True, text=True, check=True).stderr
    blocks = re.split(r"remark: Function Name: ", err)[1:]
    mangled, rows = [], []
    for b in blocks:
        mangled.append(b.split()[0])
        row = {}
        for key, pat in _FIELDS.items():
            m = re.search(r"remark:\s+" + pat + r": (\d+)", b)
            row[key] = int(m.group(1)) if m else None
        rows.append(row)
    return dict(zip(_demangle(mangled), rows))


def main(argv):
    table = kernel_resource_usage()
    print(f"{'vgpr':>5} {'sgpr':>5} {'scr':>4} {'occ':>3} {'lds':>6}  kernel")
    for name, r in sorted(table.items()):
        short = re.sub(r"^void ", "", name)
        short = re.sub(r"\(.*$", "", short)
        if argv and not any(a in short for a in argv):
            continue
        print(f"{r['vgprs']:5d} {r['sgprs']:5d} {r['scratch']:4d} {r['occupancy']:3d} {r['lds']:6d}  {short[:140]}")


if __name__ == "__main__":
    main(sys.argv[1:])
