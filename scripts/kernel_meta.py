"""Registers, LDS, scratch and spills of every kernel of one translation unit, read from the code object's metadata notes
(no GPU needed):   python scripts/kernel_meta.py conv3d [--dev]
A kernel with a scratch segment costs ~6 us of dispatch gap on either side of each launch (docs/design/08): this is the check."""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    unit = sys.argv[1]
    suffix = ".dev.o" if "--dev" in sys.argv else ".o"
    obj = os.path.join(ROOT, "densematchingbenchmark_amd", "lib", unit + suffix)
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "k.co")
        subprocess.check_call([LLVM + "/llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, obj])
        subprocess.check_call([LLVM + "/clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + fat,
                               "--output=" + co, "--unbundle"])
        notes = subprocess.run([LLVM + "/llvm-readelf", "--notes", co], capture_output=True, text=True, check=True).stdout
    rows = []
    for blk in notes.split("- .agpr_count")[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        g = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1))   # noqa: E731
        rows.append((name, g("vgpr_count"), g("sgpr_count"), g("group_segment_fixed_size"), g("private_segment_fixed_size"), g("vgpr_spill_count")))
    names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.splitlines()
    for dn, r in zip(names, rows):
        dn = re.sub(r"\(.*", "", dn).replace("dmb::", "").replace("void ", "")
        print("%-100s vgpr %3d sgpr %3d lds %6d scratch %4d spills %3d" % (dn[:100], r[1], r[2], r[3], r[4], r[5]))


if __name__ == "__main__":
    main()
