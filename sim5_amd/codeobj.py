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


def lane_moves(lib_path, name_part):
    """{mangled kernel name: (v_writelane count, v_readlane count)} for the kernels whose name contains `name_part`: the
    instructions a spilled scalar register actually costs (the metadata's sgpr_spill_count counts spill SLOTS the register
    allocator reserved, most of which it folds away again)"""
    out = {}
    tmp = tempfile.mkdtemp(prefix="s5co_")
    try:
        local = os.path.join(tmp, "lib.so")
        shutil.copy(lib_path, local)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], cwd=tmp, check=True, capture_output=True)
        for f in sorted(os.listdir(tmp)):
            if "amdgcn" not in f:
                continue
            txt = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", os.path.join(tmp, f)],
                                 capture_output=True, text=True).stdout
            cur = None
            for line in txt.splitlines():
                m = re.match(r"^[0-9a-f]+ <([^>]+)>:", line)
                if m:
                    cur = m.group(1) if (name_part in m.group(1) and not m.group(1).endswith(".kd")) else None
                    if cur:
                        out.setdefault(cur, [0, 0])
                elif cur:
                    if "v_writelane" in line:
                        out[cur][0] += 1
                    elif "v_readlane" in line:
                        out[cur][1] += 1
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return {k: tuple(v) for k, v in out.items()}
