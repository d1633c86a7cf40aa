"""Build the HIP shared library in-tree: sim5_amd/lib/libsim5gpu.so (gfx950 only).

hipcc cross-compiles without a GPU, so this also is the "does it build" check.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
# S5_VARIANT=<name> (tests/tools/ab_build.sh): an experiment build goes to lib/ab_<name>.so with its own objects and
# stamp; the library the product loads is never replaced by an experiment
_VAR = os.environ.get("S5_VARIANT", "")
OBJ = os.path.join(CSRC, "_build", "ab_" + _VAR) if _VAR else os.path.join(CSRC, "_build")
LIB = os.path.join(LIBDIR, ("ab_%s.so" % _VAR) if _VAR else "libsim5gpu.so")
LIB_RCCL = os.path.join(LIBDIR, ("ab_%s_rccl.so" % _VAR) if _VAR else "libsim5gpu_rccl.so")

# (source, object, variant): the image kernels are built in both arithmetic variants
SOURCES = [("capi_core.hip", "capi_core.o", "strict"), ("capi_batch.hip", "capi_batch.o", "strict"),
           ("capi_jobs.hip", "capi_jobs.o", "strict"), ("capi_boundary.hip", "capi_boundary.o", "strict"),
           ("k_assemble.hip", "k_assemble.o", "strict"),
           ("k_torus.hip", "k_torus_strict.o", "strict"), ("k_torus.hip", "k_torus_fast.o", "fast"),
           ("k_disk_image.hip", "k_disk_image_strict.o", "strict"), ("k_disk_image.hip", "k_disk_image_fast.o", "fast"),
           ("k_polar_image.hip", "k_polar_image_strict.o", "strict"), ("k_polar_image.hip", "k_polar_image_fast.o", "fast"),
           ("k_spectrum.hip", "k_spectrum_strict.o", "strict"), ("k_spectrum.hip", "k_spectrum_fast.o", "fast"),
           ("k_surface.hip", "k_surface_strict.o", "strict"), ("k_surface.hip", "k_surface_fast.o", "fast"),
           ("k_chain.hip", "k_chain_fast.o", "fast")]

FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17",
         "-Wall", "-Wno-unused-function", "-Wno-unused-variable"]
# The translation units are compiled with -ffp-contract=off (the reference build is x86-64 baseline, no FMA).  The STRICT
# variant keeps it that way everywhere; the FAST variant lets the compiler fuse a*b+c region by region through
# `#pragma clang fp contract(on)` at the head of its routines (s5_config.hpp: S5_FPC_MASK = 62: every region but the
# closed-form quartic, whose discriminant X = F^2 - 4E^3 agrees with the reference only when F^2 is rounded before the
# subtraction -- bisected on MI355X in round 3: fusing the quartic alone takes the worst pixel from r 6e-13 to 1.4e-7), and
# the march kernel's fast build is compiled with -ffp-contract=fast (below).
VARIANT = {"strict": ["-DS5_FAST=0", "-ffp-contract=off"], "fast": ["-DS5_FAST=1", "-ffp-contract=off"]}


def _newer(target, deps):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(d) <= t for d in deps)


def _fingerprint(files, extra):
    """Content hash of everything the library is built from: a copied tree (gpurun snapshot, checkout) keeps the
    prebuilt library whatever happened to the modification times."""
    import hashlib
    h = hashlib.sha1()
    for f in sorted(files):
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    h.update(repr(extra).encode())
    return h.hexdigest()


def build(force=False, verbose=False):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "sim5gpu.h"))
    headers.append(os.path.join(os.path.dirname(HERE), "include", "sim5gpu_rccl.h"))
    sources = sorted(set(os.path.join(CSRC, src) for (src, _, _) in SOURCES)) + [os.path.join(CSRC, "rccl_shard.hip")]
    stamp = os.path.join(LIBDIR, ("ab_%s.stamp" % _VAR) if _VAR else "build.stamp")
    # experiment flags from the environment reach the VARIANT builds only (S5_VARIANT=<name>: lib/ab_<name>.so); the library
    # the product loads is compiled with the strict / fast pair and nothing else (tests/test_capi_boundary.py checks the
    # recorded command lines)
    env = (lambda k: os.environ.get(k, "")) if _VAR else (lambda k: "")
    fp = _fingerprint(headers + sources + [os.path.abspath(__file__)],
                      (FLAGS, VARIANT, env("S5_FAST_EXTRA"), env("S5_TORUS_EXTRA"), env("S5_TORUS_FAST_EXTRA"), env("S5_SURF_FAST_EXTRA"), env("S5_IMAGE_FAST_EXTRA")))
    if not force and os.path.exists(LIB) and os.path.exists(LIB_RCCL) and os.path.exists(stamp) and open(stamp).read().strip() == fp:
        return LIB
    if os.path.exists(stamp):
        os.remove(stamp)
    objs = []
    cmds = []
    for (src, obj, variant) in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, obj)
        objs.append(o)
        extra = env("S5_FAST_EXTRA").split() if variant == "fast" else []
        if src == "k_torus.hip":
            extra = extra + env("S5_TORUS_EXTRA").split()
            if variant == "fast":
                # the march kernel of the fast variant lets the compiler fuse a*b+c (measured on MI355X: C4 job
                # 39.7 -> 37.0 ms, step counts identical to the reference's on every ray of the test sets; the
                # cancellation that rules contraction out for the image kernels is not on this path)
                extra = extra + ["-ffp-contract=fast"] + env("S5_TORUS_FAST_EXTRA").split()
        if src == "k_surface.hip" and variant == "fast":
            extra = extra + env("S5_SURF_FAST_EXTRA").split()
        if src == "k_disk_image.hip" and variant == "fast":
            extra = extra + env("S5_IMAGE_FAST_EXTRA").split()
        cmd = [hipcc] + FLAGS + VARIANT[variant] + extra + ["-c", s, "-o", o]
        # an object is reused only if it is newer than its sources AND was compiled by this very command line
        # (experiment flags from the environment must not survive in objects a later build links)
        cmdfile = o + ".cmd"
        same_cmd = os.path.exists(cmdfile) and open(cmdfile).read() == " ".join(cmd)
        if force or not same_cmd or not _newer(o, [s] + headers):
            if os.path.exists(cmdfile):
                os.remove(cmdfile)
            cmds.append(cmd)
    # the translation units are independent: compile up to 4 at a time (each hipcc peaks at ~1.5 GB)
    jobs = max(1, min(4, int(os.environ.get("S5_BUILD_JOBS", "4")), os.cpu_count() or 1))
    running = []
    def _reap(block):
        for pr, cmd in list(running):
            rc = pr.wait() if block else pr.poll()
            if rc is None:
                continue
            running.remove((pr, cmd))
            if rc != 0:
                for other, _ in running:
                    other.wait()
                raise subprocess.CalledProcessError(rc, cmd)
    for cmd in cmds:
        while len(running) >= jobs:
            _reap(False)
            if len(running) >= jobs:
                running[0][0].wait()
        if verbose:
            print(" ".join(cmd))
        running.append((subprocess.Popen(cmd), cmd))
    while running:
        _reap(True)
    for cmd in cmds:
        with open(cmd[-1] + ".cmd", "w") as fh:
            fh.write(" ".join(cmd))
    if force or cmds or not _newer(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + \
              ["-o", LIB, "-Wl,-rpath,/opt/rocm/lib", "-Wl,--no-undefined"]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    build_rccl(hipcc, force=force or bool(cmds), verbose=verbose)
    check_no_scratch(LIB)
    with open(stamp, "w") as fh:
        fh.write(fp + "\n")
    return LIB


def check_no_scratch(lib):
    """Refuse a library whose fast image kernels spill a vector register or use a private segment (a build of
    the pairing kernel with ONE spilled double gave wrong pixels in one launch mode; the cause was never pinned below the
    compiler, so such a build is an error HERE, for the in-tree library and for every experiment variant, not only in the
    test suite).  S5_ALLOW_SCRATCH=1 lets a timing experiment through."""
    if os.environ.get("S5_ALLOW_SCRATCH") == "1":
        return
    try:
        from sim5_amd.codeobj import kernel_metadata
    except ImportError:
        sys.path.insert(0, os.path.dirname(HERE))
        from sim5_amd.codeobj import kernel_metadata
    meta = kernel_metadata(lib)
    bad = {k: v for k, v in meta.items()
           if "s5f" in k and "disk_image" in k and (v.get("vgpr_spill_count", 0) or v.get("private_segment_fixed_size", 0))}
    # round 6: every whole-job kernel of the fast variant -- image, polarized image, spectrum pairs, march
    # (start, order, pool), surface search (set-up, walk, slow steps, finish) -- and the set-up kernels of both variants run
    # without a private segment and without spilled vector registers.  (Until round 6 every kernel that inlined the closed-form
    # quartic carried 32 bytes of scratch: s5_geod.hpp radial_roots.)  Left out, by name: the unpaired spectrum kernel (80 bytes:
    # the stack of the out-of-line cold routines it calls, no spill) and the batch forms of the scalar API.
    whole_job = ("disk_image", "torus_start", "torus_order", "torus_pool", "surface_setup", "surface_walk", "surface_slow", "surface_finish",
                 "disk_spectrum_fast_kernelILb1E", "geodesic_chain_kernel")
    for k, v in meta.items():
        if "s5f" in k and any(w in k for w in whole_job) and (v.get("vgpr_spill_count", 0) or v.get("private_segment_fixed_size", 0)):
            bad[k] = v
        if "_ZN2s5" in k and ("torus_start" in k or "surface_setup" in k) and v.get("private_segment_fixed_size", 0):
            bad[k] = v
    if bad:
        os.remove(lib)
        raise RuntimeError("build refused: whole-job kernels with spilled VGPRs / scratch: %r" % bad)


def build_rccl(hipcc, force=False, verbose=False):
    """libsim5gpu_rccl.so: the multi-GPU form of the image job (include/sim5gpu_rccl.h) -- host code on top of the base
    library's C-ABI and librccl, a library of its own so that libsim5gpu.so keeps no RCCL dependency."""
    src = os.path.join(CSRC, "rccl_shard.hip")
    deps = [src, os.path.join(os.path.dirname(HERE), "include", "sim5gpu_rccl.h"), os.path.join(os.path.dirname(HERE), "include", "sim5gpu.h"), LIB]
    if not force and _newer(LIB_RCCL, deps):
        return LIB_RCCL
    cmd = [hipcc, "--offload-arch=gfx950", "-O2", "-fPIC", "-std=c++17", "-Wall", "-shared", src, "-o", LIB_RCCL,
           "-L" + LIBDIR, "-l:" + os.path.basename(LIB), "-L/opt/rocm/lib", "-lrccl",
           "-Wl,-rpath,$ORIGIN", "-Wl,-rpath,/opt/rocm/lib", "-Wl,--no-undefined"]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return LIB_RCCL


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
