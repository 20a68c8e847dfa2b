#!/usr/bin/env python3
"""Register / scratch / LDS table of every kernel of cvo_kernels.hip, from the compiler's own
metadata (no GPU needed):

    python tools/isa_resources.py [out.txt]

compiles cvo-rgbd_amd/csrc/cvo_kernels.hip with the Makefile's flags + -save-temps into a scratch
directory and reads the amdhsa.kernels notes of the gfx950 assembly."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "cvo-rgbd_amd", "csrc")
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-slp-vectorize", "-Wno-invalid-offsetof",
         "-I" + os.path.join(ROOT, "include"), "-I" + SRC, "--offload-arch=gfx950", "-mllvm", "-amdgpu-kernarg-preload-count=8", "-save-temps"]


def table(asm):
    rows, cur = [], None
    keys = (".sgpr_count", ".sgpr_spill_count", ".vgpr_count", ".vgpr_spill_count", ".private_segment_fixed_size",
            ".group_segment_fixed_size")
    for line in asm.splitlines():
        s = line.strip()
        if s.startswith("- .agpr_count") or s.startswith("- .args"):
            if cur:
                rows.append(cur)
            cur = {}
        m = re.match(r"-?\s*(\.[a-z_]+):\s*(\S+)", s)
        if cur is not None and m:
            if m.group(1) in keys:
                cur[m.group(1)] = int(m.group(2))
            elif m.group(1) == ".name":
                cur["name"] = m.group(2)
    if cur:
        rows.append(cur)
    return [r for r in rows if "name" in r], keys


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else None
    with tempfile.TemporaryDirectory() as tmp:
        subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + ["-c", os.path.join(SRC, "cvo_kernels.hip"), "-o",
                                                           os.path.join(tmp, "k.o")], cwd=tmp, check=True)
        asm = [f for f in os.listdir(tmp) if f.endswith(".s") and "gfx950" in f]
        text = open(os.path.join(tmp, asm[0])).read()
    rows, keys = table(text)
    mfma = len(re.findall(r"^\s*v_mfma_", text, re.M))
    swaps = len(re.findall(r"^\s*v_permlane\d+_swap", text, re.M))
    lines = ["# ISA resources of cvo_kernels.hip (hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -mllvm -amdgpu-kernarg-preload-count=8 -save-temps)",
             "# %d v_mfma_*, %d v_permlane*_swap in the file" % (mfma, swaps),
             "# kernel | sgpr | sgpr spilled (to vector lanes, no memory) | vgpr | vgpr spilled | scratch bytes | static LDS bytes"]
    for r in rows:
        name = re.sub(r"^_ZN7cvo_dev", "", r["name"])
        lines.append(" | ".join([name] + [str(r.get(k, 0)) for k in keys]))
    body = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(body)
    sys.stdout.write(body)


if __name__ == "__main__":
    main()
