"""Per-kernel register / spill / LDS report of a HIP source compiled for gfx950 (no GPU needed):
    python tools/kernel_regs.py understanding_flow_robustness_amd/csrc/igemm.hip [filter]
"""
import os
import re
import subprocess
import sys
import tempfile

src = os.path.abspath(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
with tempfile.TemporaryDirectory() as d:
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-function",
                    "-c", src, "-o", os.path.join(d, "x.o"), "-save-temps=obj"], check=True, stderr=subprocess.DEVNULL, cwd=d)
    s = open(next(os.path.join(d, f) for f in os.listdir(d) if f.endswith("gfx950.s"))).read()
keys = (".vgpr_count", ".agpr_count", ".vgpr_spill_count", ".sgpr_spill_count", ".private_segment_fixed_size", ".group_segment_fixed_size")
for m in re.finditer(r"- \.agpr_count:.*?\.wavefront_size", s, re.S):
    body = m.group(0)
    name = re.search(r"\.name:\s+(\S+)", body).group(1)
    dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().replace("(anonymous namespace)::", "")
    if flt and flt not in dn:
        continue
    num = lambda k: re.search(re.escape(k) + r":\s+(\d+)", body).group(1)
    vals = " ".join(k[1:].replace("_count", "").replace("_fixed_size", "") + "=" + num(k) for k in keys)
    print(dn[:80].ljust(80), vals)
