"""Register / spill / LDS table of one HIP translation unit (no GPU needed).
    python scripts/kernel_resources.py torchcde_amd/csrc/dopri5_mlp_adjoint.hip [name-filter]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from torchcde_amd import _lib  # noqa: E402

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
flags = [f for f in _lib.HIPCC_FLAGS if f != "-shared"] + _lib.EXTRA_FLAGS.get(os.path.basename(src), [])
out = subprocess.run([_lib._hipcc()] + flags + ["-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"],
                     stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True).stdout
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"remark:\s+([^:]+): (\S+)", line)
    if not m:
        if "error" in line:
            print(line)
        continue
    key, val = m.group(1).strip(), m.group(2)
    if key == "Function Name":
        cur = {"name": val}
        rows.append(cur)
    elif cur is not None:
        cur[key] = val
names = subprocess.run(["c++filt"] + [r["name"] for r in rows], stdout=subprocess.PIPE, text=True).stdout.splitlines()
print("%-6s %-6s %-7s %-8s %-4s %-8s %s" % ("VGPR", "AGPR", "spill", "scratch", "occ", "LDS", "kernel"))
for r, n in zip(rows, names):
    n = re.sub(r"\(.*", "", n).replace("void cde::", "")
    if flt in n:
        print("%-6s %-6s %-7s %-8s %-4s %-8s %s" % (r.get("VGPRs"), r.get("AGPRs"), r.get("VGPRs Spill"), r.get("ScratchSize [bytes/lane]"),
                                                      r.get("Occupancy [waves/SIMD]"), r.get("LDS Size [bytes/block]"), n))
