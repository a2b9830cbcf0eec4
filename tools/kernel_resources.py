"""Per-kernel register / LDS / scratch usage from the gfx950 assembly of one translation unit.

    python tools/kernel_resources.py batch3dmot_amd/csrc/b3d_pose.hip [filter]
"""
import os, re, subprocess, sys, tempfile

src = os.path.abspath(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
root = os.path.dirname(os.path.dirname(os.path.dirname(src)))
with tempfile.TemporaryDirectory() as d:
    subprocess.run(["hipcc", "-std=c++20", "-O3", "--offload-arch=gfx950", "-c", src, "-o", "x.o", "--save-temps",
                    "-I" + os.path.join(root, "include")], cwd=d, check=True, stderr=subprocess.DEVNULL)
    asm = [f for f in os.listdir(d) if f.endswith("gfx950.s")][0]
    text = open(os.path.join(d, asm)).read()
meta = text[text.index("amdhsa.kernels:"):]
for blk in meta.split("  - .agpr_count:")[1:]:
    blk = ".agpr_count:" + blk
    f = dict(re.findall(r"\.(\w+):\s+(\S+)", blk))
    name = subprocess.run(["c++filt", f["name"]], capture_output=True, text=True).stdout.strip()
    name = name.replace("b3d::", "").replace("MPDims<48, 32, 0, 96, 64, 96, 64, 96, 64>", "P").replace("MPDims<96, 64, 64, 256, 128, 192, 128, 192, 128>", "C")
    if flt and flt not in name:
        continue
    print(f"vgpr {f['vgpr_count']:>4s} agpr {f['agpr_count']:>4s} spill {f.get('vgpr_spill_count','0'):>4s} scratch {f['private_segment_fixed_size']:>6s} "
          f"lds {f['group_segment_fixed_size']:>7s} sgpr {f['sgpr_count']:>4s}  {name.split('(')[0][:110]}")
