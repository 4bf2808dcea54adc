"""Basic-block view of the gfx950 code of one kernel (no GPU needed): where the MFMAs, the spills (scratch), the barriers
and the full vmcnt waits sit, and how large the code is.  Used to keep hot loops inside the 64 KB instruction cache and free
of scratch traffic.
    python scripts/asm_blocks.py torchcde_amd/csrc/dopri5_mlp_adjoint.hip 'dopri5_mlp_adjoint_attempt<3, 1, 8, 8, true>'"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from torchcde_amd import _lib  # noqa: E402


def main():
    src, want = os.path.abspath(sys.argv[1]), sys.argv[2]
    flags = [f for f in _lib.HIPCC_FLAGS if f != "-shared"] + _lib.EXTRA_FLAGS.get(os.path.basename(src), [])
    with tempfile.TemporaryDirectory(prefix="cde_asm_") as tmp:
        subprocess.run([_lib._hipcc()] + flags + ["-c", src, "-o", "x.o", "--save-temps"], cwd=tmp, stdout=subprocess.DEVNULL,
                       stderr=subprocess.DEVNULL)
        text = open(glob.glob(os.path.join(tmp, "*gfx950.s"))[0]).read()
    parts = re.split(r"\n(_Z\w+):", text)
    names = subprocess.run(["c++filt"] + parts[1::2], stdout=subprocess.PIPE, text=True).stdout.splitlines()
    for name, body in zip(names, parts[2::2]):
        if want not in name:
            continue
        body = body.split(".Lfunc_end")[0]
        blocks = re.split(r"\n(\.LBB\d+_\d+):", body)
        items = [("entry", blocks[0])] + list(zip(blocks[1::2], blocks[2::2]))
        total = 0
        print(re.sub(r"\(.*", "", name))
        for label, block in items:
            ins = [ln for ln in block.split("\n") if ln.startswith("\t") and not ln.startswith("\t.") and not ln.startswith("\t;")]
            total += len(ins)
            mfma = sum("v_mfma" in ln for ln in ins)
            scratch = sum("scratch_" in ln for ln in ins)
            if mfma or scratch > 3 or len(ins) > 200:
                print("  %-12s %5d instr  mfma %4d  scratch %3d  accvgpr %4d  barrier %2d  vmcnt(0) %2d  lds %3d  global %3d" % (
                    label, len(ins), mfma, scratch, sum("accvgpr" in ln for ln in ins), sum("s_barrier" in ln for ln in ins),
                    sum("vmcnt(0)" in ln for ln in ins), sum("\tds_" in ln for ln in ins), sum("\tglobal_" in ln for ln in ins)))
        print("  total %d instructions" % total)


if __name__ == "__main__":
    main()
