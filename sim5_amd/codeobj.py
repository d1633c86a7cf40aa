"""Register and scratch figures of the kernels inside the built library, read from the code objects' metadata
(llvm-objdump --offloading + llvm-readelf --notes from /opt/rocm/lib/llvm/bin).  Used by the build check that keeps
register spills out of the image kernels and by the notes in DESIGN.md; nothing on the product path imports it."""
import os
import re
import shutil
import subprocess
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
FIELDS = ("private_segment_fixed_size", "sgpr_count", "sgpr_spill_count", "vgpr_count", "vgpr_spill_count", "group_segment_fixed_size")


def kernel_metadata(lib_path):
    """{mangled kernel name: {field: int}} over every gfx950 code object bundled in the library"""
    out = {}
    tmp = tempfile.mkdtemp(prefix="s5co_")
    try:
        local = os.path.join(tmp, "lib.so")
        shutil.copy(lib_path, local)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], cwd=tmp, check=True, capture_output=True)
        for f in sorted(os.listdir(tmp)):
            if "amdgcn" not in f:
                continue
            txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(tmp, f)], capture_output=True, text=True).stdout
            for block in re.split(r"\n\s+- \.agpr_count:", txt)[1:]:
                name = re.search(r"\.name:\s+(\S+)", block)
                if not name:
                    continue
                rec = {}
                for k in FIELDS:
                    m = re.search(r"\.%s:\s+(\d+)" % k, block)
                    if m:
                        rec[k] = int(m.group(1))
                out[name.group(1)] = rec
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return out
