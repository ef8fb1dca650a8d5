"""Registers, LDS and scratch of every kernel in a built HIP library, read from its gfx950 code objects (no GPU needed).

    python tools/kernel_resources.py                 # eesen_amd/lib/libeesen_hip.so -> profiles/kernel_resources.md
    python tools/kernel_resources.py LIB.so          # any other library (tests/native/libfake_rccl.so), table on stdout

How: llvm-objcopy dumps the .hip_fatbin section (with -fno-gpu-rdc: one clang offload bundle per translation unit, concatenated),
clang-offload-bundler unbundles each for hipv4-amdgcn-amd-amdhsa--gfx950, llvm-readelf --notes prints the AMDGPU metadata
(.vgpr_count, .agpr_count, .sgpr_count, .group_segment_fixed_size, .private_segment_fixed_size, spill counts,
.max_flat_workgroup_size).  These are the numbers hipFuncGetAttributes reports at run time and the ones the schedule decisions of
the library rest on (co-residency of the persistent recurrence grids with side-stream GEMMs and with RCCL's all-reduce
workgroups: DESIGN.md section 7); tests/test_kernel_resources.py pins them.
"""
from __future__ import annotations

import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
FIELDS = {"vgprs": ".vgpr_count", "agprs": ".agpr_count", "sgprs": ".sgpr_count", "lds": ".group_segment_fixed_size",
          "scratch": ".private_segment_fixed_size", "vgpr_spill": ".vgpr_spill_count", "sgpr_spill": ".sgpr_spill_count",
          "max_threads": ".max_flat_workgroup_size"}


def _tool(name: str) -> str:
    p = os.path.join(LLVM, name)
    return p if os.path.exists(p) else (shutil.which(name) or name)


def demangle(names):
    r = subprocess.run([shutil.which("c++filt") or _tool("llvm-cxxfilt")], input="\n".join(names), capture_output=True, text=True, check=True)
    out = []
    for d in r.stdout.splitlines():
        d = re.sub(r"\(anonymous namespace\)::", "", d)
        d = re.sub(r"^void\s+", "", d)
        d = re.sub(r"\(.*$", "", d)            # drop the parameter list
        d = d.replace("eesen::", "").replace(" ", "")
        d = re.sub(r"\(bool\)1|true", "true", d)
        d = re.sub(r"\(bool\)0|false", "false", d)
        out.append(d)
    return out


def kernels(lib: str):
    """[{name (demangled, no parameter list), mangled, vgprs, agprs, sgprs, lds, scratch, vgpr_spill, sgpr_spill, max_threads}]"""
    rows = []
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fat.bin")
        subprocess.run([_tool("llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, lib], check=True, capture_output=True)
        blob = open(fat, "rb").read()
        offs = [m.start() for m in re.finditer(MAGIC, blob)]
        for i, o in enumerate(offs):
            piece, co = os.path.join(td, f"b{i}.bin"), os.path.join(td, f"d{i}.co")
            open(piece, "wb").write(blob[o: offs[i + 1] if i + 1 < len(offs) else len(blob)])
            subprocess.run([_tool("clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + piece, "--targets=" + TARGET, "--output=" + co],
                           check=True, capture_output=True)
            notes = subprocess.run([_tool("llvm-readelf"), "--notes", co], capture_output=True, text=True, check=True).stdout
            for block in re.split(r"\n\s+- \.agpr_count:", "\n" + notes)[1:]:
                block = "  - .agpr_count:" + block
                row = {}
                for k, f in FIELDS.items():
                    m = re.search(re.escape(f) + r":\s+(\d+)", block)
                    row[k] = int(m.group(1)) if m else 0
                m = re.search(r"\.name:\s+(\S+)", block)
                if not m:
                    continue
                row["mangled"] = m.group(1)
                rows.append(row)
    for row, d in zip(rows, demangle([r["mangled"] for r in rows])):
        row["name"] = d
    return sorted(rows, key=lambda r: r["name"])


def markdown(rows, title: str) -> str:
    out = [f"# {title}", "",
           "Read from the built library's gfx950 code objects by `tools/kernel_resources.py` (llvm-readelf --notes); `vgprs` is the code object's",
           "`.vgpr_count` per lane of a 64-wide wave (allocated in blocks of 8; a SIMD holds 512 per lane), `lds` the static",
           "`.group_segment_fixed_size` in bytes.  Pinned by `tests/test_kernel_resources.py` (budgets in its `BUDGETS` table).", "",
           "| kernel | threads | vgprs | agprs | sgprs | lds | scratch | spills (v/s) |", "|---|---:|---:|---:|---:|---:|---:|---|"]
    for r in rows:
        out.append(f"| `{r['name']}` | {r['max_threads']} | {r['vgprs']} | {r['agprs']} | {r['sgprs']} | {r['lds']} | {r['scratch']} | {r['vgpr_spill']}/{r['sgpr_spill']} |")
    return "\n".join(out) + "\n"


def main():
    if len(sys.argv) > 1:
        print(markdown(kernels(sys.argv[1]), os.path.basename(sys.argv[1])))
        return
    lib = os.path.join(ROOT, "eesen_amd", "lib", "libeesen_hip.so")
    rows = kernels(lib)
    fake = os.path.join(ROOT, "tests", "native", "libfake_rccl.so")
    text = markdown(rows, "Kernel resources of libeesen_hip.so (gfx950)")
    if os.path.exists(fake):
        text += "\n" + markdown(kernels(fake), "tests/native/libfake_rccl.so (the stand-in; `rccl_shaped` = ncclDevKernel_Generic_*'s footprint)").replace("# ", "## ", 1)
    open(os.path.join(ROOT, "profiles", "kernel_resources.md"), "w").write(text)
    print(f"{len(rows)} kernels -> profiles/kernel_resources.md")


if __name__ == "__main__":
    main()
