"""ctypes binding of include/sim5gpu_rccl.h (libsim5gpu_rccl.so): the multi-GPU form of the thin-disk image job in C --
one process per GPU, mirrored stripe pairs, one RCCL gather per image, the image assembled on rank 0.  Nothing here
computes rays; without the library the import fails (there is no CPU fallback)."""
import ctypes as C
import os

from . import capi as _c

# the multi-GPU library that was linked against the base library in use: an experiment variant sim5_amd/lib/ab_<name>.so
# (SIM5GPU_LIB, tests/tools/ab_build.sh) has its own ab_<name>_rccl.so next to it -- the in-tree pair is never mixed with it
_base = os.path.basename(_c.LIB_PATH)
_pair = (_base[:-3] + "_rccl.so") if (_base.startswith("ab_") and _base.endswith(".so")) else "libsim5gpu_rccl.so"
LIB_PATH = os.environ.get("SIM5GPU_RCCL_LIB") or os.path.join(os.path.dirname(_c.LIB_PATH), _pair)
if not os.path.exists(LIB_PATH):
    raise ImportError("sim5_amd: %s is missing -- build it with `python -m sim5_amd.build`" % LIB_PATH)
_lib = C.CDLL(LIB_PATH)
_lib.sim5gpu_rccl_last_error.restype = C.c_char_p
ID_BYTES = 128
STRIPE_ROWS = 64
VP, I = C.c_void_p, C.c_int


class Sim5GpuRcclError(RuntimeError):
    pass


def _check(rc, what):
    if rc != 0:
        raise Sim5GpuRcclError("%s failed (status %d): %s" % (what, rc, _lib.sim5gpu_rccl_last_error().decode(errors="replace")))


def unique_id():
    buf = C.create_string_buffer(ID_BYTES)
    _check(_lib.sim5gpu_rccl_unique_id(buf), "sim5gpu_rccl_unique_id")
    return buf.raw


def comm_create(id_bytes, rank, world):
    assert len(id_bytes) == ID_BYTES
    comm = VP()
    _check(_lib.sim5gpu_rccl_comm_create(C.c_char_p(id_bytes), I(rank), I(world), C.byref(comm)), "sim5gpu_rccl_comm_create")
    return comm


def comm_destroy(comm):
    _check(_lib.sim5gpu_rccl_comm_destroy(comm), "sim5gpu_rccl_comm_destroy")


def shard_plan(desc, rank, world, dealt_rows=0):
    """(rows this rank traces per image, (band_y0, band_y1), job description of its share): host arithmetic, no GPU"""
    rows, b0, b1 = I(0), I(0), I(0)
    share = _c.ImageDesc()
    _check(_lib.sim5gpu_shard_plan(C.byref(desc), I(rank), I(world), I(dealt_rows), C.byref(rows), C.byref(b0), C.byref(b1),
                                   C.byref(share)), "sim5gpu_shard_plan")
    return rows.value, (b0.value, b1.value), share


class Shard:
    def __init__(self, comm, rank, world, nx, ny, dealt_rows=0):
        self.ptr = VP()
        _check(_lib.sim5gpu_shard_create(C.byref(self.ptr), comm, I(rank), I(world), I(nx), I(ny), I(dealt_rows)), "sim5gpu_shard_create")

    def begin(self, desc, d_image_f=None, d_image_g=None, stream=None):
        _check(_lib.sim5gpu_shard_image_begin(self.ptr, C.byref(desc), VP(d_image_f or 0), VP(d_image_g or 0), VP(stream or 0)),
               "sim5gpu_shard_image_begin")

    def end(self, stream=None):
        _check(_lib.sim5gpu_shard_image_end(self.ptr, VP(stream or 0)), "sim5gpu_shard_image_end")

    def image(self, desc, d_image_f=None, d_image_g=None, stream=None):
        _check(_lib.sim5gpu_disk_image_sharded(self.ptr, C.byref(desc), VP(d_image_f or 0), VP(d_image_g or 0), VP(stream or 0)),
               "sim5gpu_disk_image_sharded")

    def destroy(self):
        if self.ptr:
            _lib.sim5gpu_shard_destroy(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass
