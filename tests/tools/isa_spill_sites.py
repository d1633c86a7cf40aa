"""HERE (no GPU): where the production image kernel keeps scalar registers in vector lanes.  Compiles k_disk_image.hip (fast
variant) to assembly with region marks (-DS5_ISA_MARKS: comments at the head of the cold re-trace and of the owed-flux pass)
and lists every v_writelane / v_readlane of disk_image_jobs_kernel<true> with the region it falls in: the hot path -- set-up,
the R_F evaluation, ladder, crossings, g-factor, flux table, stores, three class instantiations of it -- holds a handful
(HOT_MAX below; round 4: 4 writes and 6 reads over the three instantiations), the rest sits in the cold re-trace."""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = os.path.join(ROOT, "sim5_amd", "csrc", "k_disk_image.hip")
out = os.path.join(tempfile.mkdtemp(prefix="s5isa_"), "k.s")
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-DS5_FAST=1", "-ffp-contract=off",
       "-DS5_ISA_MARKS", "-S", "--cuda-device-only", src, "-o", out]
subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
lines = open(out).read().splitlines()
name = "_ZN3s5f22disk_image_jobs_kernelILb1EEEvN5s5abi7JobListE"
a = next(i for i, l in enumerate(lines) if l.startswith(name + ":"))
b = next(i for i in range(a, len(lines)) if lines[i].startswith(".Lfunc_end"))       # (a kernel has several s_endpgm)
body = lines[a:b]
region, hot = "hot path", 0
counts = {}
for l in body:
    m = re.search(r"S5MARK (.*)", l)
    if m:
        region = m.group(1).strip()
    if "v_writelane" in l or "v_readlane" in l:
        counts[region] = counts.get(region, 0) + 1
print("disk_image_jobs_kernel<true>: %d instructions; lane moves by region: %r" % (sum(1 for l in body if l.startswith("\t") and not l.startswith("\t;")), counts))
HOT_MAX = 12
sys.exit(1 if counts.get("hot path", 0) > HOT_MAX else 0)
