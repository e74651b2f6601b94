"""Per-kernel register and scratch figures from the device assembly of rt_kernels.hip.

    hipcc ... --cuda-device-only -S raytracer_amd/csrc/rt_kernels.hip -o /tmp/rt_kernels.s
    python profiles/isa_report.py /tmp/rt_kernels.s [substring]
"""
import re
import subprocess
import sys

text = open(sys.argv[1]).read()
want = sys.argv[2] if len(sys.argv) > 2 else ""
meta = text[text.index("amdhsa.kernels:"):]
for blk in meta.split("  - .agpr_count:")[1:]:
    name = re.search(r"\.name:\s+(\S+)", blk).group(1)
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dem = dem.split("(")[0].replace("void ", "")
    if want not in dem:
        continue
    g = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1))
    print("%-46s vgpr %3d spill %3d sgpr %3d scratch %5d B lds %5d B" % (dem, g("vgpr_count"), g("vgpr_spill_count"), g("sgpr_count"),
                                                                         g("private_segment_fixed_size"), g("group_segment_fixed_size")))
