#!/usr/bin/env python3
"""Per-kernel register / scratch usage of the built library, read from the AMDGPU metadata notes of the gfx950 code objects inside
difashion_amd/csrc/*.o (the device ELF of each object: .hip_fatbin -> clang-offload-bundler --unbundle -> llvm-readelf --notes).

  python scripts/kernel_resources.py            # table of every kernel that spills or uses scratch
  python scripts/kernel_resources.py --all      # every kernel

tests/test_cabi_cpu.py::test_no_product_kernel_spills uses kernel_table() to fail the build check on any spilled VGPR."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"


def demangle(names):
    try:
        out = subprocess.run([os.path.join(LLVM, "llvm-cxxfilt")] + names, capture_output=True, text=True, check=True).stdout.splitlines()
        return dict(zip(names, out))
    except Exception:
        return {n: n for n in names}


def object_kernels(obj):
    """[(mangled name, {field: int})] of one host object file with an embedded gfx950 code object."""
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "dev.co")
        r = subprocess.run([os.path.join(LLVM, "llvm-objcopy"), f"--dump-section=.hip_fatbin={fat}", obj], capture_output=True, text=True)
        if r.returncode != 0 or not os.path.exists(fat):
            return []
        subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={fat}", f"--targets={TARGET}",
                        f"--output={co}"], check=True, capture_output=True)
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True, check=True).stdout
    out, cur = [], None
    for line in notes.splitlines():
        m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)$", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2).strip()
        if k == "agpr_count" or (k == "args" and cur is None):
            pass
        if line.lstrip().startswith("- .") and k in ("agpr_count", "args"):      # first key of a kernel record (keys are sorted)
            cur = {}
            out.append(cur)
        if cur is None:
            continue
        if k == "name" and "name" not in cur and not v.startswith("'") and "hidden" not in v:
            pass
        if k in ("vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size",
                 "group_segment_fixed_size", "max_flat_workgroup_size"):
            cur[k] = int(v)
        elif k == "symbol":
            cur["symbol"] = v.strip("'")
    return [(c["symbol"][:-3] if c.get("symbol", "").endswith(".kd") else c.get("symbol", "?"), c) for c in out if "vgpr_count" in c]


def kernel_table(csrc=None):
    """{demangled kernel name: fields} over every object of the product library; ``fields["mangled"]`` is the Itanium symbol (what the
    tests match on: it does not depend on llvm-cxxfilt being installed)."""
    csrc = csrc or os.path.join(ROOT, "difashion_amd", "csrc")
    rows = []
    for f in sorted(os.listdir(csrc)):
        if f.endswith(".o"):
            rows += [(f, n, c) for n, c in object_kernels(os.path.join(csrc, f))]
    names = demangle([n for _, n, _ in rows])
    return {names[n]: dict(c, object=f, mangled=n) for f, n, c in rows}


if __name__ == "__main__":
    tab = kernel_table()
    show_all = "--all" in sys.argv
    print(f"{'vgpr':>5} {'agpr':>5} {'spill':>5} {'scratch':>7} {'lds':>7}  kernel")
    for name, c in sorted(tab.items(), key=lambda kv: (-kv[1].get("vgpr_spill_count", 0), kv[0])):
        if show_all or c.get("vgpr_spill_count", 0) or c.get("private_segment_fixed_size", 0):
            print(f"{c.get('vgpr_count', 0):5d} {c.get('agpr_count', 0):5d} {c.get('vgpr_spill_count', 0):5d} {c.get('private_segment_fixed_size', 0):7d} "
                  f"{c.get('group_segment_fixed_size', 0):7d}  {name}  [{c['object']}]")
    print(f"{len(tab)} kernels, {sum(1 for c in tab.values() if c.get('vgpr_spill_count', 0))} with spilled VGPRs")
