"""sim5_amd/csrc/s5_x87.hpp -- the integer restatement of the x87 double-extended statement group in the reference's
geodesic_priv_T_roots (ref: /root/reference/src/sim5kerr-geod.c:1125-1131) -- compiled for the host (it uses no float
instruction of the device, so the bits are the device's) and held to (i) the CPU's real long double, operation by operation,
on random operands, and (ii) the m2m / m2p fields of records the UNMODIFIED reference makes here (oracle/_ref), including the
rays with l = 0 whose range test `m2p >= 1.0` only those roundings decide."""
import os
import platform
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "libsim5ref.so")


@pytest.mark.skipif(platform.machine() != "x86_64", reason="needs the x87 long double of an x86-64 host")
def test_x87_emulation_against_long_double_and_reference(tmp_path):
    exe = str(tmp_path / "x87_check")
    subprocess.run(["g++", "-O1", "-ffp-contract=off", os.path.join(ROOT, "tests", "c", "x87_check.cpp"), "-o", exe, "-ldl"], check=True)
    args = [exe, "400000"] + ([REF] if os.path.exists(REF) else [])
    p = subprocess.run(args, capture_output=True, text=True, timeout=300)
    print(p.stdout)
    assert p.returncode == 0 and "ok " in p.stdout, p.stdout + p.stderr
    if os.path.exists(REF):
        assert "reference records" in p.stdout
